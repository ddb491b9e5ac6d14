"""Data-parallel training on the GPU with two ranks (BASELINE.json configs[2] in miniature, the loop of
reference/train.py:109-119 sharded by batch): two fresh processes, both on the one visible GPU, train on the halves
of a batch through ``unet_amd.Trainer`` - bucket-by-bucket backward, one asynchronous all-reduce per bucket on the
communication stream while the next bucket computes, AdamW behind the last one - and must follow a one-process run on
the whole batch.  The exchange is gloo (RCCL refuses two ranks on one device; the RCCL bucket path itself runs in
tests/test_dp_rccl_gpu.py), for which the reducer stages each bucket through pinned host memory: a correctness
rehearsal of the multi-rank step, not a model of its speed."""
import os
import socket
import subprocess
import sys

import pytest
import torch

pytestmark = pytest.mark.gpu
HERE = os.path.dirname(os.path.abspath(__file__))


def _spawn(world, tmp_path, tag):
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    outs = [str(tmp_path / f"{tag}_{r}.pt") for r in range(world)]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    procs = [subprocess.Popen([sys.executable, "-u", os.path.join(HERE, "dp_child.py"), str(r), str(world), str(port), outs[r]],
                              env=env, stdout=subprocess.PIPE, stderr=subprocess.STDOUT, text=True) for r in range(world)]
    logs = []
    try:
        for p in procs:
            logs.append(p.communicate(timeout=115)[0])
    except subprocess.TimeoutExpired:
        for p in procs:
            p.kill()
        raise AssertionError("a rank did not finish in time:\n" + "\n".join(logs))
    finally:
        for p in procs:
            if p.poll() is None:
                p.kill()
    for r, p in enumerate(procs):
        if p.returncode != 0:
            low = logs[r].lower()
            if "gloo" in low and ("not supported" in low or "no backend" in low):
                pytest.skip(f"device tensors not supported by this gloo build:\n{logs[r][-800:]}")
            others = "\n".join(f"--- rank {q} ({procs[q].returncode}) ---\n{logs[q][-2000:]}" for q in range(world) if q != r)
            raise AssertionError(f"rank {r} failed ({p.returncode}):\n{logs[r][-3000:]}\n{others}")
    return [torch.load(o, weights_only=True) for o in outs]


def test_two_ranks_follow_the_full_batch_run(dev, tmp_path):
    ref = _spawn(1, tmp_path, "full")[0]
    r0, r1 = _spawn(2, tmp_path, "dp")
    assert r0["world"] == 2 and r1["world"] == 2 and r0["buckets"] == 4 and ref["buckets"] == 0
    assert torch.equal(r0["params"], r1["params"])                 # replicas stay identical, bit for bit
    assert torch.equal(r0["grads"], r1["grads"])
    # averaged shard gradients = full-batch gradient (Dice and BCE are means over samples / pixels)
    err = float((r0["grads"] - ref["grads"]).abs().max() / ref["grads"].abs().max())
    assert err < 1e-4, err
    # Adam turns rounding-level gradients (biases in front of an instance norm) into +-lr steps, so the
    # parameters are only compared loosely: three steps of lr = 1e-3
    assert float((r0["params"] - ref["params"]).abs().max()) < 3.5e-3
