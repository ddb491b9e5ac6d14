"""Lifetime and resource behaviour of the engine on the GPU: workspaces that come and go, a bounded plan cache
under sliding-window inference over many image sizes, finite parameter buffers on the autograd path, and the
optimizer's checkpoint format."""
import gc

import pytest
import torch

import unet_amd
from test_unet_gpu import disks

pytestmark = pytest.mark.gpu


def _step(plan, flat, x, dlog, ws, seed=5):
    logits = torch.empty_like(x)
    g = torch.zeros_like(flat)
    plan.forward(flat, x, logits, ws, True, seed)
    plan.backward(flat, x, dlog, g, ws, True, seed)
    torch.cuda.synchronize()
    return logits.clone(), g.clone()


def test_a_recycled_workspace_block_needs_no_preparation(dev):
    """The engine keeps nothing in a workspace between calls (descriptor tables travel as kernel arguments): a
    block that was freed, overwritten by other tensors and handed out again at the same address works at once."""
    torch.manual_seed(2)
    m = unet_amd.UNet(2, 1, 1, (16, 32, 64), (2, 2), num_res_units=2, dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
    x = torch.rand(2, 1, 64, 64, device=dev)
    dlog = torch.randn(2, 1, 64, 64, device=dev) * 1e-3
    plan = m._plan_for(x, True)
    flat = m._ensure_flat(dev)
    ws = torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=dev)
    addr = ws.data_ptr()
    y0, g0 = _step(plan, flat, x, dlog, ws)
    del ws
    junk = torch.full((plan.workspace_bytes // 4,), float("nan"), dtype=torch.float32, device=dev)   # scribble over the block
    same_block = junk.data_ptr() == addr
    del junk
    ws2 = torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=dev)
    y1, g1 = _step(plan, flat, x, dlog, ws2)
    assert torch.equal(y0, y1) and torch.equal(g0, g1)
    # and a second workspace at another address, with the first still alive, gives the same bits as well
    ws3 = torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=dev)
    assert ws3.data_ptr() != ws2.data_ptr()
    y2, g2 = _step(plan, flat, x, dlog, ws3)
    assert torch.equal(y0, y2) and torch.equal(g0, g2)
    print("recycled block had the same address:", same_block)


def test_sliding_window_over_many_image_sizes_keeps_memory_bounded(dev):
    torch.manual_seed(3)
    m = unet_amd.UNet(2, 1, 1, (16, 32, 64), (2, 2), num_res_units=2, compute_dtype=torch.bfloat16).to(dev).eval()
    sizes = [(96 + 8 * i, 160 - 8 * (i % 7)) for i in range(20)] + [(64, 64), (70, 50), (333, 97)]
    peak = 0
    with torch.no_grad():
        for k, (h, w) in enumerate(sizes):
            x = torch.rand(1 + k % 3, 1, h, w, device=dev)
            y = unet_amd.sliding_window_inference(x, (64, 64), 4, m)
            assert y.shape == x.shape and bool(torch.isfinite(y).all())
            del x, y
            if k == 4:
                gc.collect(); torch.cuda.synchronize()
                base = torch.cuda.memory_allocated(dev)
            if k > 4:
                gc.collect(); torch.cuda.synchronize()
                peak = max(peak, torch.cuda.memory_allocated(dev))
    assert len(m._plans) <= m.MAX_PLANS
    batch_sizes = sorted({k[0] for k in m._plans})
    assert all(b & (b - 1) == 0 or b == 288 for b in batch_sizes), batch_sizes      # powers of two (or the maximum) only
    assert peak <= 2 * base + (64 << 20), (base, peak)


def test_autograd_path_keeps_the_flat_buffers_finite(dev):
    torch.manual_seed(4)
    m = unet_amd.UNet(2, 1, 1, (16, 32, 64), (2, 2), num_res_units=2, compute_dtype=torch.float32).to(dev).train()
    opt = unet_amd.AdamW(m.parameters(), lr=1e-3, model=m)
    loss = unet_amd.DiceCELoss(sigmoid=True)
    x = torch.rand(2, 1, 32, 32, device=dev)
    t = disks(2, (32, 32), 3).to(dev)
    poison = torch.full((1 << 22,), float("nan"), device=dev)      # make stale allocator blocks hold NaN
    del poison
    for _ in range(3):
        opt.zero_grad()
        loss(m(x), t).backward()
        assert bool(torch.isfinite(m._last_grad_flat).all())
        opt.step()
    torch.cuda.synchronize()
    assert bool(torch.isfinite(m.flat_parameters()).all())
    assert bool(torch.isfinite(opt._m).all()) and bool(torch.isfinite(opt._v).all())


def test_optimizer_state_round_trips_in_torch_format(dev):
    def make():
        torch.manual_seed(6)
        m = unet_amd.UNet(2, 1, 1, (16, 32), (2,), num_res_units=2, compute_dtype=torch.float32).to(dev).train()
        return m, unet_amd.AdamW(m.parameters(), lr=2e-3, weight_decay=1e-2, model=m)

    loss = unet_amd.DiceCELoss(sigmoid=True)
    x = torch.rand(2, 1, 32, 32, device=dev)
    t = disks(2, (32, 32), 8).to(dev)

    def step(m, opt):
        opt.zero_grad()
        loss(m(x), t).backward()
        opt.step()

    m1, o1 = make()
    for _ in range(3):
        step(m1, o1)
    sd_model = {k: v.clone() for k, v in m1.state_dict().items()}
    sd_opt = o1.state_dict()
    assert len(sd_opt["state"]) == 21 and float(sd_opt["state"][0]["step"]) == 3.0
    step(m1, o1)
    # resume in a fresh model / optimizer
    m2, o2 = make()
    m2.load_state_dict(sd_model)
    o2.load_state_dict(sd_opt)
    step(m2, o2)
    torch.cuda.synchronize()
    assert torch.equal(m1.flat_parameters(), m2.flat_parameters())
    # the same file drives torch.optim.AdamW on the same parameters to the same place (to rounding)
    m3, _ = make()
    m3.load_state_dict(sd_model)
    ref = [torch.nn.Parameter(p.detach().clone()) for p in m3.parameters()]
    o3 = torch.optim.AdamW(ref, lr=2e-3, weight_decay=1e-2)
    o3.load_state_dict(sd_opt)
    m3.zero_grad()
    loss(m3(x), t).backward()
    for r, p in zip(ref, m3.parameters()):
        r.grad = p.grad.detach().clone()
    o3.step()
    for r, p in zip(ref, m1.parameters()):
        assert float((r.detach() - p.detach()).abs().max()) < 1e-6


def test_optimizer_state_loaded_on_the_cpu_follows_the_model_to_the_gpu(dev):
    """A checkpoint loaded while the model still lives on the CPU must keep its moments when the model moves:
    one more step has to land where an uninterrupted run lands, bit for bit."""
    def make(device):
        torch.manual_seed(6)
        m = unet_amd.UNet(2, 1, 1, (16, 32), (2,), num_res_units=2, compute_dtype=torch.float32).to(device).train()
        return m, unet_amd.AdamW(m.parameters(), lr=2e-3, weight_decay=1e-2, model=m)

    loss = unet_amd.DiceCELoss(sigmoid=True)
    x = torch.rand(2, 1, 32, 32, device=dev)
    t = disks(2, (32, 32), 8).to(dev)

    def step(m, opt):
        opt.zero_grad()
        loss(m(x), t).backward()
        opt.step()

    m1, o1 = make(dev)
    for _ in range(3):
        step(m1, o1)
    sd_model = {k: v.detach().cpu().clone() for k, v in m1.state_dict().items()}
    sd_opt = o1.state_dict()
    sd_opt["state"] = {i: {k: v.cpu() for k, v in s.items()} for i, s in sd_opt["state"].items()}
    step(m1, o1)
    m2, o2 = make("cpu")
    m2.load_state_dict(sd_model)
    o2.load_state_dict(sd_opt)
    assert o2._m is not None and not o2._m.is_cuda and float(o2._m.abs().sum()) > 0
    m2.to(dev)
    step(m2, o2)
    torch.cuda.synchronize()
    assert o2._m.is_cuda and o2._step == 4
    assert torch.equal(m1.flat_parameters(), m2.flat_parameters())


def test_a_training_step_replays_from_a_captured_graph(dev):
    """A caller may capture the whole step (forward, loss, backward, AdamW: one library call) into a HIP graph.  The
    engine's side stream joins the capture through its fork / join events; the start-flag forks of small workloads
    (stream value waits, which a capture does not take) give way to events while the stream is capturing.  One
    replay of a captured second step must leave the parameters where an eager second step leaves them."""
    x = torch.rand(4, 1, 64, 64, device=dev)
    t = (torch.rand(4, 1, 64, 64, device=dev) > 0.6).float()

    def make():
        torch.manual_seed(11)
        m = unet_amd.UNet(2, 1, 1, (16, 32, 64), (2, 2), num_res_units=2, dropout=0.0, compute_dtype=torch.float32).to(dev).train()
        opt = unet_amd.AdamW(m.parameters(), lr=1e-3, weight_decay=1e-5, model=m)
        return m, unet_amd.Trainer(m, unet_amd.DiceCELoss(sigmoid=True), opt)

    m_e, tr_e = make()
    tr_e.step(x, t); tr_e.step(x, t)
    torch.cuda.synchronize()
    want = m_e.flat_parameters().detach().clone()

    m_g, tr_g = make()
    tr_g.step(x, t)                      # eager first step: buffers, plan and side stream exist before the capture
    torch.cuda.synchronize()
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        tr_g.step(x, t)
    graph.replay()
    torch.cuda.synchronize()
    got = m_g.flat_parameters().detach()
    assert torch.isfinite(got).all()
    assert torch.equal(got, want)


def test_bucket_calls_with_a_ready_stream_give_the_whole_pass_bit_for_bit(dev):
    """``mi_unet_backward_bucket``: the four bucket calls, each handing the wait for its weight gradients to another
    stream (as the data-parallel trainer hands it to the communication stream), leave exactly the gradients of the
    one-call backward pass once the two streams have been joined."""
    torch.manual_seed(4)
    m = unet_amd.UNet(2, 1, 1, (16, 32, 64, 128), (2, 2, 2), num_res_units=2, dropout=0.1, compute_dtype=torch.bfloat16).to(dev).train()
    x = torch.rand(4, 1, 64, 64, device=dev)
    dlog = torch.randn(4, 1, 64, 64, device=dev) * 1e-3
    plan = m._plan_for(x, True)
    flat = m._ensure_flat(dev)
    ws = torch.empty(plan.workspace_bytes, dtype=torch.uint8, device=dev)
    logits = torch.empty_like(x)
    plan.forward(flat, x, logits, ws, True, 9)
    g_all = torch.zeros_like(flat)
    plan.backward(flat, x, dlog, g_all, ws, True, 9)
    torch.cuda.synchronize()
    other = torch.cuda.Stream(dev)
    g_b = torch.zeros_like(flat)
    nbuckets = len(plan.bucket_ranges())
    for b in range(nbuckets):
        plan.backward(flat, x, dlog, g_b, ws, True, 9, bucket=b, ready_stream=other)
    torch.cuda.current_stream(dev).wait_stream(other)
    torch.cuda.synchronize()
    assert nbuckets == 4
    assert torch.equal(g_b, g_all) and torch.isfinite(g_b).all()
