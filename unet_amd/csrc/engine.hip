// Network-level engine: turns a MONAI-style residual U-Net configuration into a fixed list
// of kernel launches (forward and backward), owns the workspace layout and runs the lists on
// a HIP stream.  Pure host logic plus launches; no device allocation, no synchronisation.
//
// Data layout decisions (all in the caller-provided workspace):
//  * hidden activations are channels-last in the compute dtype; skip concatenation is free:
//    the encoder output and the decoder output of a level are written side by side into
//    one buffer whose pitch is the concatenated channel count;
//  * the stride-2 encoder convolution and its stride-2 residual convolution share one
//    launch (2*C output channels, same input tile);
//  * convolutions that feed an instance norm skip their bias (it cancels exactly) and emit
//    the per-(image, channel) sums the norm needs from their epilogue;
//  * weight gradients are accumulated per workgroup, written as slabs and reduced in a
//    fixed order (deterministic), one reduction launch per gradient bucket.
#include <math.h>
#include <stdio.h>

#include <algorithm>
#include <chrono>
#include <map>
#include <string>
#include <type_traits>
#include <vector>

#include "common.h"
#include "geom.h"
#include "kparams.h"
#include "lossparams.h"
#include "../../include/unet_mi355.h"

static thread_local std::string g_err;
void mi_set_error(const std::string& s) { g_err = s; }
extern "C" const char* mi_last_error(void) { return g_err.c_str(); }
extern "C" int mi_version(void) { return 100; }

namespace {

enum StepKind {
    K_IGEMM, K_WGRAD, K_NORM_FWD, K_NORM_BWD, K_C1_FWD, K_C1_T, K_C1_WGRAD, K_STENCIL, K_STENCIL_WGRAD, K_ADD,
    K_MEMSET, K_REDUCE, K_PACK, K_KFINISH,
    K_KFINISH_NORM      // finishing pass of a split convolution (ig) and the norm forward behind it (nm, relocated by rel_nm) in one launch
};
enum Base { B_ARENA, B_PARAMS, B_GRADS, B_XIN, B_LOGITS, B_DLOGITS, B_NBASE };
constexpr int MI_NFLAG = 64;     // start flags per backward pass (forks released by a kernel's start instead of an event)

struct Reloc { size_t field_off; int base; };

struct AddArgs { char* dst; int dpitch, dcoff; const char* src; int spitch, scoff, c; long long npix; };
struct MemsetArgs { char* ptr; size_t bytes; };
struct TabArgs { int first, count; };   // range of descriptors a table step launches

struct Step {
    int kind = 0, dtype = 0, bucket = 0;
    double work = 0; int work_is_flops = 0; double bytes = 0;
    std::string layer, kernel;
    IgemmParams ig; WgradParams wg; NormParams nm; C1Params c1; StencilParams stn;
    AddArgs add; MemsetArgs ms; TabArgs tab;
    std::vector<Reloc> rel;
    std::vector<Reloc> rel_nm;   // K_KFINISH_NORM: relocations of nm (rel belongs to ig)
    double total_ms = 0; long long launches = 0;
    int wait_pack = 0;     // forward: this step is the first to read packed weights (waits for the first packing launch)
    int join_side = 0;     // forward: this step is the first to read weights of the second packing launch (joins the side stream)
    Step() {
        memset(&ig, 0, sizeof(ig)); memset(&wg, 0, sizeof(wg)); memset(&nm, 0, sizeof(nm));
        memset(&c1, 0, sizeof(c1)); memset(&stn, 0, sizeof(stn)); memset(&add, 0, sizeof(add));
        memset(&ms, 0, sizeof(ms)); memset(&tab, 0, sizeof(tab));
    }
};

#define REL(step, ps, field, base, off)                                                               \
    do {                                                                                              \
        (ps).field = (decltype((ps).field))(uintptr_t)(off);                                          \
        (step).rel.push_back(Reloc{offsetof(std::remove_reference<decltype(ps)>::type, field), base}); \
    } while (0)

bool bases_ok(const Step& s, char* const* bases) {
    for (const Reloc& r : s.rel)
        if (!bases[r.base]) return false;
    for (const Reloc& r : s.rel_nm)
        if (!bases[r.base]) return false;
    return true;
}

template <class P>
P resolve_with(const std::vector<Reloc>& rel, const P& src, char* const* bases) {
    P q = src;
    for (const Reloc& r : rel) {
        char** f = reinterpret_cast<char**>(reinterpret_cast<char*>(&q) + r.field_off);
        *f = bases[r.base] + reinterpret_cast<uintptr_t>(*f);
    }
    return q;
}
template <class P>
P resolve(const Step& s, const P& src, char* const* bases) { return resolve_with(s.rel, src, bases); }

#define REL_TO(vec, ps, field, base, off)                                                            \
    do {                                                                                              \
        (ps).field = (decltype((ps).field))(uintptr_t)(off);                                          \
        (vec).push_back(Reloc{offsetof(std::remove_reference<decltype(ps)>::type, field), base});     \
    } while (0)

struct Ten {  // a channels-last tensor inside the arena
    size_t off = 0; int pitch = 0; int coff = 0; int base = B_ARENA;
};

struct ParamRec {
    std::string name; int ndim; int shape[5]; long long numel; long long offset; int bucket;
};

struct ProfEvent { hipEvent_t a, b; int backward, step; };

}  // namespace

struct mi_unet {
    mi_unet_cfg cfg;
    int L, R, nd, dt, esz, nb;
    int KK;  // taps of a 3^nd kernel
    std::vector<ParamRec> params;
    std::map<std::string, int> pindex;
    long long nfloats = 0;
    long long bucket_off[MI_NUM_BUCKETS], bucket_cnt[MI_NUM_BUCKETS];
    std::vector<Step> fwd, bwd;
    size_t arena = 0;
    // descriptor tables (pointer fields hold offsets until resolved)
    std::vector<PackDesc> packs; std::vector<int> pack_is_bwd;
    std::vector<ReduceDesc> reduces; std::vector<int> reduce_has_b;
    size_t off_stats = 0, stats_bytes = 0, off_bstats = 0, bstats_bytes = 0;
    size_t off_flags = 0;                 // MI_NFLAG start flags, 64 bytes apart, inside the backward statistics block (cleared with it)
    double flops_fwd = 0, flops_bwd = 0;
    int n_fwd_packs = 0;
    std::vector<mi_norm_info> norms;      // norm blocks of the forward pass in execution order
    // side stream for the weight-gradient launches of the backward pass (they depend only on tensors that are
    // final when they are issued and are joined in front of each bucket's slab reduction)
    hipStream_t side = nullptr;
    std::vector<hipEvent_t> side_ev;      // one "inputs ready" event per weight-gradient step, then one join event
    hipEvent_t join_ev = nullptr;
    std::vector<hipEvent_t> late_ev;      // "weight gradients of this bucket are enqueued" marks for reductions that run on the main stream
    hipEvent_t pack_ev = nullptr;         // forward: the first packing launch (weights of the first levels) is done
    int use_side = -1;                    // -1: undecided, 0: off, 1: on
    int can_wait_value = -1;              // hipStreamWaitValue32 usable on this device (-1: not asked yet)
    unsigned flag_seq = 0;                // value the start flags of the current backward call take (never 0, changes every call)
    const void* flag_arena = nullptr;     // workspace whose flag words have been cleared for this plan
    // profiling
    bool prof = false;
    std::vector<ProfEvent> events;
    int rd[9][3]; long long rp[9];
    // MI_HOST_TIMING=1: wall time the host spends inside mi_unet_train_step (enqueue only), reported when the plan goes
    double host_ns = 0; long long host_calls = 0;
};

namespace {

size_t align_up(size_t v, size_t a) { return (v + a - 1) / a * a; }

struct Builder {
    mi_unet& u;
    explicit Builder(mi_unet& uu) : u(uu) {}
    int layer_id = 0;
    int cur_bucket = 0;

    size_t alloc(size_t bytes) {
        size_t o = u.arena;
        u.arena = align_up(u.arena + bytes, 256);
        return o;
    }
    Ten new_tensor(int res, int pitch, int esz) {
        Ten t; t.off = alloc((size_t)u.nb * u.rp[res] * pitch * esz); t.pitch = pitch; t.coff = 0; return t;
    }
    long long poff(const std::string& name) { return u.params[u.pindex.at(name)].offset; }

    ConvShape shape(int transposed, int ksize, int stride, int cin, int cout, int res_in) {
        ConvShape s; s.ndim = u.nd; s.transposed = transposed; s.ksize = ksize; s.stride = stride;
        s.cin = cin; s.cout = cout; s.nb = u.nb;
        for (int d = 0; d < 3; ++d) s.in_dims[d] = u.rd[res_in][d];
        return s;
    }

    // ---- packed weights -----------------------------------------------------------------------
    // returns arena offset of the packed tensor [ntaps][Ntot][Ktot]
    size_t packed_alloc(int ntaps, int Ntot, int Ktot) { return alloc((size_t)ntaps * Ntot * Ktot * u.esz); }
    void add_pack(bool bwd, size_t dst_off, const std::string& wname, int N, int K, int ntaps, const int* kidx,
                  int sn, int sk, int Ntot, int Ktot, int n_off, int k_off) {
        PackDesc d; memset(&d, 0, sizeof(d));
        d.src = (const float*)(uintptr_t)(poff(wname) * sizeof(float));
        d.dst = (void*)(uintptr_t)dst_off;
        d.N = N; d.K = K; d.ntaps = ntaps; d.sn = sn; d.sk = sk; d.Ntot = Ntot; d.Ktot = Ktot;
        d.n_off = n_off; d.k_off = k_off;
        for (int i = 0; i < ntaps; ++i) d.kidx[i] = (signed char)kidx[i];
        u.packs.push_back(d); u.pack_is_bwd.push_back(bwd ? 1 : 0);
    }

    // ---- reductions ----------------------------------------------------------------------------
    void add_reduce(size_t partial_off, size_t pbias_off, int nslab, int ntaps, int Ftot, int S, int f_off, int Fcnt,
                    int KKdst, const int* kidx, const std::string& wname, const std::string& bname) {
        ReduceDesc d; memset(&d, 0, sizeof(d));
        d.partial = (const float*)(uintptr_t)partial_off; d.partial_bias = (const float*)(uintptr_t)pbias_off;
        d.nslab = nslab; d.ntaps = ntaps; d.Ftot = Ftot; d.S = S; d.f_off = f_off; d.Fcnt = Fcnt; d.KK = KKdst;
        for (int i = 0; i < ntaps; ++i) d.kidx[i] = (signed char)(kidx ? kidx[i] : i);
        d.dst_w = (float*)(uintptr_t)(poff(wname) * sizeof(float));
        int hasb = 0;
        if (!bname.empty()) { d.dst_b = (float*)(uintptr_t)(poff(bname) * sizeof(float)); hasb = 1; }
        u.reduces.push_back(d); u.reduce_has_b.push_back(hasb);
        group_of_reduce.push_back(n_reduce_steps);
    }
    // scalar copy (PReLU slope gradient) or zero fill (nslab = 0)
    void add_scalar_reduce(size_t src_off, int nslab, const std::string& name, int count = 1) {
        ReduceDesc d; memset(&d, 0, sizeof(d));
        d.partial = (const float*)(uintptr_t)src_off; d.partial_bias = nullptr;
        d.nslab = nslab; d.ntaps = 1; d.Ftot = count; d.S = 1; d.f_off = 0; d.Fcnt = count; d.KK = 1; d.kidx[0] = 0;
        d.dst_w = (float*)(uintptr_t)(poff(name) * sizeof(float));
        u.reduces.push_back(d); u.reduce_has_b.push_back(0);
        group_of_reduce.push_back(n_reduce_steps);
    }
    // a reduction descriptor belongs to the next K_REDUCE step pushed after it
    std::vector<int> group_of_reduce;
    int n_reduce_steps = 0;
    void push_reduce_step(const std::string& layer) {
        Step& s = push(u.bwd, K_REDUCE, layer); s.kernel = "reduce_kernel";
        ++n_reduce_steps;
    }

    // ---- steps ---------------------------------------------------------------------------------
    Step& push(std::vector<Step>& list, int kind, const std::string& layer) {
        list.emplace_back();
        Step& s = list.back();
        s.kind = kind; s.dtype = u.dt; s.bucket = cur_bucket; s.layer = layer;
        return s;
    }

    // generic igemm step. in1 optional (c_in0 < K). stats_off = (size_t)-1: none.
    bool add_igemm(std::vector<Step>& list, const std::string& layer, const ConvShape& cs, int dir, const Ten& in0,
                   int in0_c, const Ten* in1, int K, int N, size_t w_off, const char* bias_name, int bias_c0,
                   const Ten& out, const Ten* addend, size_t stats_off, int stats_c, int nstat, int kidx_out[MI_MAX_TAPS],
                   int* ntaps_out, const Ten* out1 = nullptr, int out0_c = 0) {
        Step& s = push(list, K_IGEMM, layer);
        IgemmParams& p = s.ig;
        if (!igemm_geometry(cs, dir, p, kidx_out, ntaps_out)) { mi_set_error("igemm geometry: odd size under stride 2"); return false; }
        REL(s, p, in0, in0.base, in0.off);
        p.in0_pitch = in0.pitch; p.in0_coff = in0.coff; p.in0_c = in0_c;
        if (in1) { REL(s, p, in1, in1->base, in1->off); p.in1_pitch = in1->pitch; p.in1_coff = in1->coff; }
        p.K = K; p.N = N;
        REL(s, p, w, B_ARENA, w_off);
        if (bias_name) { REL(s, p, bias, B_PARAMS, poff(bias_name) * sizeof(float)); p.bias_c0 = bias_c0; }
        REL(s, p, out, out.base, out.off); p.out_pitch = out.pitch; p.out_coff = out.coff;
        p.out0_c = N;
        if (out1) { REL(s, p, out1, out1->base, out1->off); p.out1_pitch = out1->pitch; p.out1_coff = out1->coff; p.out0_c = out0_c; }
        if (addend) { REL(s, p, addend, addend->base, addend->off); p.add_pitch = addend->pitch; p.add_coff = addend->coff; }
        if (stats_off != (size_t)-1) { REL(s, p, stats, B_ARENA, stats_off); p.stats_c = stats_c; p.nstat = nstat; }
        int cps, nfrag;
        igemm_config(u.dt, p, &cps, &nfrag);
        char kb[96];
        snprintf(kb, sizeof(kb), "igemm_kernel<%s,%d,%d>", u.dt ? "BF16" : "F32", cps, nfrag);
        if (conv2d_dma_enabled() && conv2d_dma_applicable(u.dt, p))
            conv2d_dma_label(u.dt, p, kb, sizeof(kb));
        else if (conv2d_dma_enabled() && conv3d_k16_applicable(u.dt, p))
            snprintf(kb, sizeof(kb), "conv3d_k16_kernel");
        else
            igemm_dma_label(u.dt, p, kb, sizeof(kb));
        s.kernel = kb;
        double bpix = (double)p.bd[0] * p.bd[1] * p.bd[2] * u.nb;
        s.work = 2.0 * bpix * (double)(*ntaps_out) * K * N; s.work_is_flops = 1;
        {
            const double ipix = (double)p.id[0] * p.id[1] * p.id[2] * u.nb, opix = (double)p.od[0] * p.od[1] * p.od[2] * u.nb;
            s.bytes = (ipix * K + opix * N * (addend ? 2.0 : 1.0)) * u.esz + (double)(*ntaps_out) * K * N * u.esz;
        }
        if (&list == &u.fwd) u.flops_fwd += s.work; else u.flops_bwd += s.work;
        // Too few workgroups for the chip and several contraction chunks (deep levels of small batches, 3-D
        // patches): split the chunks over more workgroups and finish the layer in a second, tiny launch.
        {
            long long wgs = 0; int nkc = 1;
            igemm_parallelism(u.dt, p, &wgs, &nkc);
            int ks = 1;
            // a forward 1 x 1 (x 1) convolution is a handful of contraction steps: splitting it buys a second launch
            // and nothing else (in the backward pass its finishing pass carries the norm backward, so it stays split)
            const bool one_tap_fwd = *ntaps_out == 1 && &list == &u.fwd && !getenv("MI_SPLIT_1TAP_FWD");
            if (!getenv("MI_NO_KSPLIT") && !one_tap_fwd && wgs < 256 && nkc >= 2 && N % 4 == 0 && N / 4 <= 256) {
                // workgroups aimed at: 512 in 3-D; 64 in 2-D, where a split layer leaves the asynchronously staged
                // kernels for the generic one and only the smallest layers gain (measured: batch 8 x 256^2 0.943 ->
                // 0.881 ms per step at 64 instead of 512; 96^3 patches, with the finishing pass inside the norm
                // launches: 1.700 / 1.578 / 1.539 / 1.524 / 1.526 ms at 64 / 128 / 256 / 512 / 1024)
                static const int kst_env = getenv("MI_KSPLIT_TARGET") ? atoi(getenv("MI_KSPLIT_TARGET")) : 0;
                const int kst = kst_env > 0 ? kst_env : (u.nd == 3 ? 512 : 64);
                ks = (int)((kst + wgs - 1) / wgs);
                if (ks > nkc) ks = nkc;
                if (ks > 8) ks = 8;
            }
            if (ks > 1) {
                const size_t npo = (size_t)u.nb * p.od[0] * p.od[1] * p.od[2];
                const size_t kp = alloc((size_t)ks * npo * N * sizeof(float));
                p.ksplit = ks;
                REL(s, p, kpart, B_ARENA, kp);
                {   // a split layer leaves the compile-time 2-D kernels: name the kernel that serves it now
                    char kb2[96];
                    snprintf(kb2, sizeof(kb2), "igemm_kernel<%s,%d,%d>", u.dt ? "BF16" : "F32", cps, nfrag);
                    igemm_dma_label(u.dt, p, kb2, sizeof(kb2));
                    s.kernel = kb2;
                }
                const std::string lname = s.layer;
                const Step main_step = s;              // copy: push() may move the vector's storage
                Step& f = push(list, K_KFINISH, lname + " (finish)");
                f.ig = main_step.ig; f.rel = main_step.rel; f.dtype = main_step.dtype;
                f.kernel = u.dt ? "ksplit_finish_kernel<BF16>" : "ksplit_finish_kernel<F32>";
                f.work = 0; f.work_is_flops = 0;
                f.bytes = (double)npo * N * (ks * 4.0 + u.esz);
            }
        }
        return true;
    }

    int wgrad_target = 0;      // workgroups wanted for the next weight gradients (0: the default)
    // weight-gradient step; returns slab offsets through the reduce descriptors the caller adds
    void add_wgrad(const std::string& layer, const ConvShape& cs, const Ten& f0, int f0_c, const Ten* f1, int F,
                   const Ten& sten, int S, size_t* partial_off, size_t* pbias_off, int* nslab, int* ntaps) {
        Step& s = push(u.bwd, K_WGRAD, layer);
        WgradParams& p = s.wg;
        wgrad_geometry(cs, u.esz, p);
        p.F = F; p.S = S; p.f0_c = f0_c;
        REL(s, p, f0, f0.base, f0.off); p.f0_pitch = f0.pitch; p.f0_coff = f0.coff;
        if (f1) { REL(s, p, f1, f1->base, f1->off); p.f1_pitch = f1->pitch; p.f1_coff = f1->coff; }
        REL(s, p, s, sten.base, sten.off); p.s_pitch = sten.pitch; p.s_coff = sten.coff;
        int ff, sf, wf, wsn;
        wgrad_choose_split(u.dt, p, wgrad_target);
        wgrad_config(u.dt, p, &ff, &sf, &wf, &wsn);
        const int slabs = wgrad_slabs(u.dt, p);
        const size_t po = alloc((size_t)slabs * p.ntaps_total * F * S * sizeof(float));
        const size_t pb = alloc((size_t)slabs * F * sizeof(float));
        REL(s, p, partial, B_ARENA, po);
        REL(s, p, partial_bias, B_ARENA, pb);
        *partial_off = po; *pbias_off = pb; *nslab = slabs; *ntaps = p.ntaps_total;
        char kb[96];
        snprintf(kb, sizeof(kb), "wgrad_kernel<%s,%d,%d,%d,%d,%d>", u.dt ? "BF16" : "F32", p.ntaps_total / p.ntapg, ff, sf, wf, wsn);
        wgrad_thick_label(u.dt, p, ff, sf, wf, wsn, kb, sizeof(kb));
        s.kernel = kb;
        double bpix = (double)p.bd[0] * p.bd[1] * p.bd[2] * u.nb;
        s.work = 2.0 * bpix * p.ntaps_total * (double)F * S; s.work_is_flops = 1;
        s.bytes = (bpix * F + (double)p.sd[0] * p.sd[1] * p.sd[2] * u.nb * S) * u.esz + (double)p.ntaps_total * F * S * 4.0;
        u.flops_bwd += s.work;
    }

    void add_norm_fwd(const std::string& layer, int res, int c, int dt, const Ten& raw, size_t stats_off,
                      const std::string& alpha, const Ten* resid, const Ten& out, int lid) {
        // A contraction-split convolution of a small image that produced `raw` a step or three ago: its finishing
        // pass takes the norm with it (one launch instead of two).  With a residual operand the finishing pass must be
        // the last step, so that nothing between it and here can be the residual's producer, and must not have a second
        // destination (the residual half of a fused two-output convolution is written by other workgroups of that launch).
        Step* fin = nullptr;
        for (size_t back = 1; back <= 3 && back <= u.fwd.size(); ++back) {
            Step& t = u.fwd[u.fwd.size() - back];
            if (t.kind != K_KFINISH || back == u.fwd.size()) continue;
            const Step& cv = u.fwd[u.fwd.size() - back - 1];
            if (cv.kind != K_IGEMM || cv.ig.kpart != t.ig.kpart) continue;
            if (raw.base == B_ARENA && (size_t)(uintptr_t)t.ig.out == raw.off && t.ig.out_pitch == raw.pitch && t.ig.out_coff == raw.coff &&
                (size_t)(uintptr_t)t.ig.stats == stats_off && t.dtype == dt && (!resid || (back == 1 && !t.ig.out1)) &&
                finish_norm_ok(t.ig, c, (int)u.rp[res])) {
                fin = &t;
                break;
            }
        }
        Step& s = fin ? *fin : push(u.fwd, K_NORM_FWD, layer);
        std::vector<Reloc>& rel = fin ? s.rel_nm : s.rel;
        s.dtype = dt;
        NormParams& p = s.nm;
        REL_TO(rel, p, x, raw.base, raw.off); p.x_pitch = raw.pitch; p.x_coff = raw.coff;
        if (resid) { REL_TO(rel, p, res, resid->base, resid->off); p.res_pitch = resid->pitch; p.res_coff = resid->coff; }
        REL_TO(rel, p, out, out.base, out.off); p.out_pitch = out.pitch; p.out_coff = out.coff;
        REL_TO(rel, p, stats, B_ARENA, stats_off);
        REL_TO(rel, p, alpha, B_PARAMS, poff(alpha) * sizeof(float));
        p.nb = u.nb; p.npix = (int)u.rp[res]; p.c = c; p.eps = 1e-5f; p.layer = (unsigned)lid;
        const int esz = dt ? 2 : 4;
        const double nbytes = (double)u.nb * u.rp[res] * c * esz * (resid ? 3.0 : 2.0);
        if (fin) {
            // the convolution in front of its finishing pass writes the partial sums in the layout this launch reads
            Step& conv = *(fin - 1);
            conv.ig.kpart_blocked = 1; s.ig.kpart_blocked = 1;
            s.kind = K_KFINISH_NORM;
            s.layer += " + " + layer.substr(layer.rfind(' ') == std::string::npos ? 0 : layer.rfind(' ') + 1);
            s.kernel = dt ? "finish_norm_kernel<BF16>" : "finish_norm_kernel<F32>";
            s.bytes += nbytes;
        } else {
            s.kernel = c == 1 ? "norm1_fwd_kernel" : (dt ? "norm_fwd_kernel<BF16>" : "norm_fwd_kernel<F32>");
            s.work = nbytes;
        }
        record_norm(res, c, dt, raw, stats_off, lid);
    }
    void record_norm(int res, int c, int dt, const Ten& raw, size_t stats_off, int lid) {
        mi_norm_info ni; memset(&ni, 0, sizeof(ni));
        ni.raw_offset = raw.base == B_ARENA ? (int64_t)raw.off : -1; ni.pitch = raw.pitch; ni.coff = raw.coff; ni.c = c;
        ni.dtype = dt; ni.npix = u.rp[res]; ni.stats_offset = (int64_t)stats_off; ni.layer = (uint32_t)lid;
        u.norms.push_back(ni);
    }

    void add_norm_bwd(const std::string& layer, int res, int c, int dt, const Ten& dy, const Ten& raw, size_t stats_off,
                      const std::string& alpha, const Ten& dx, int lid) {
        // A contraction-split data-gradient convolution of a small image that produced `dy` a few steps ago (a weight
        // gradient or a gradient reduction may sit in between): its finishing pass takes this norm backward with it.
        Step* fin = nullptr;
        for (size_t back = 1; back <= 4 && back < u.bwd.size(); ++back) {
            Step& t = u.bwd[u.bwd.size() - back];
            if (t.kind == K_WGRAD || t.kind == K_C1_WGRAD || t.kind == K_STENCIL_WGRAD || t.kind == K_REDUCE) continue;
            if (t.kind != K_KFINISH) break;          // any other step may be the one that completes dy
            const Step& cv = u.bwd[u.bwd.size() - back - 1];
            if (cv.kind == K_IGEMM && cv.ig.kpart == t.ig.kpart && dy.base == B_ARENA && (size_t)(uintptr_t)t.ig.out == dy.off &&
                t.ig.out_pitch == dy.pitch && t.ig.out_coff == dy.coff && t.dtype == dt &&
                finish_norm_bwd_ok(t.ig, dt, c, (int)u.rp[res], u.nb))
                fin = &t;
            break;
        }
        Step& s = fin ? *fin : push(u.bwd, K_NORM_BWD, layer);
        std::vector<Reloc>& rel = fin ? s.rel_nm : s.rel;
        s.dtype = dt;
        NormParams& p = s.nm;
        REL_TO(rel, p, x, raw.base, raw.off); p.x_pitch = raw.pitch; p.x_coff = raw.coff;
        REL_TO(rel, p, dy, dy.base, dy.off); p.dy_pitch = dy.pitch; p.dy_coff = dy.coff;
        REL_TO(rel, p, out, dx.base, dx.off); p.out_pitch = dx.pitch; p.out_coff = dx.coff;
        REL_TO(rel, p, stats, B_ARENA, stats_off);
        const size_t bs = bstats_alloc((size_t)u.nb * MI_NSLOT * c * 2 * sizeof(double));
        const int nda = u.nb * norm_bwd_reduce_blocks((int)u.rp[res], c);   // one slope-gradient partial per workgroup
        const size_t da = bstats_alloc((size_t)nda * sizeof(float));
        REL_TO(rel, p, bstats, B_ARENA, bs);
        REL_TO(rel, p, dalpha, B_ARENA, da);
        p.dalpha_slots = 1;
        REL_TO(rel, p, alpha, B_PARAMS, poff(alpha) * sizeof(float));
        p.nb = u.nb; p.npix = (int)u.rp[res]; p.c = c; p.eps = 1e-5f; p.layer = (unsigned)lid;
        const int esz = dt ? 2 : 4;
        const double nbytes = (double)u.nb * u.rp[res] * c * esz * 3.0;   // algorithmic: x and dy read, dx written (the kernels read x and dy twice)
        if (fin) {
            Step& conv = *(fin - 1);
            conv.ig.kpart_blocked = 1; s.ig.kpart_blocked = 1;
            s.kind = K_KFINISH_NORM;
            s.layer += " + " + layer.substr(layer.find(' ') == std::string::npos ? 0 : layer.find(' ') + 1);
            s.kernel = dt ? "finish_norm_bwd_kernel<BF16>" : "finish_norm_bwd_kernel<F32>";
            s.bytes += nbytes;
        } else {
            s.kernel = c == 1 ? "norm1_bwd_kernels" : (dt ? "norm_bwd_kernels<BF16>" : "norm_bwd_kernels<F32>");
            s.work = nbytes;
        }
        add_scalar_reduce(da, nda, alpha);
    }

    // statistics regions are carved from two contiguous blocks so one memset clears each
    size_t stats_cursor = 0, bstats_cursor = 0;
    size_t stats_alloc(size_t bytes) { size_t o = u.off_stats + stats_cursor; stats_cursor = align_up(stats_cursor + bytes, 16); return o; }
    size_t bstats_alloc(size_t bytes) { size_t o = u.off_bstats + bstats_cursor; bstats_cursor = align_up(bstats_cursor + bytes, 16); return o; }
};

// enumerate parameters in MONAI's registration (state_dict) order
void enum_params(mi_unet& u) {
    const int L = u.L, R = u.R, nd = u.nd;
    const int* c = u.cfg.channels;
    auto add = [&](const std::string& name, std::vector<int> shape, int bucket) {
        ParamRec r; r.name = name; r.ndim = (int)shape.size(); r.numel = 1;
        for (int i = 0; i < 5; ++i) r.shape[i] = i < (int)shape.size() ? shape[i] : 1;
        for (int v : shape) r.numel *= v;
        r.offset = -1; r.bucket = bucket;
        u.pindex[name] = (int)u.params.size();
        u.params.push_back(r);
    };
    auto wshape = [&](int a, int b, int k) {
        std::vector<int> s = {a, b};
        for (int i = 0; i < nd; ++i) s.push_back(k);
        return s;
    };
    // recursive blocks
    struct Rec {
        static void block(mi_unet& u, const std::string& P, int l, decltype(add)& add, decltype(wshape)& wshape) {
            const int L = u.L, R = u.R; const int* c = u.cfg.channels;
            const int cin = l == 0 ? u.cfg.in_channels : c[l - 1];
            const int cout = c[l];
            // down residual unit at P.0 (stride 2)
            res_unit(u, P + ".0", cin, cout, 3, R, false, l == 0 ? 3 : 2, add, wshape);
            if (l < L - 2) block(u, P + ".1.submodule", l + 1, add, wshape);
            else res_unit(u, P + ".1.submodule", c[L - 2], c[L - 1], (c[L - 2] != c[L - 1]) ? 1 : 0, R, false, 1, add, wshape);
            // up layer at P.2
            const int upc = (l < L - 2) ? 2 * cout : c[L - 2] + c[L - 1];
            const int outc = l == 0 ? u.cfg.out_channels : c[l - 1];
            const int bk = 0;
            add(P + ".2.0.conv.weight", wshape(upc, outc, 3), bk);
            add(P + ".2.0.conv.bias", {outc}, bk);
            add(P + ".2.0.adn.A.weight", {1}, bk);
            add(P + ".2.1.conv.unit0.conv.weight", wshape(outc, outc, 3), bk);
            add(P + ".2.1.conv.unit0.conv.bias", {outc}, bk);
            if (l != 0) add(P + ".2.1.conv.unit0.adn.A.weight", {1}, bk);
        }
        // rk: residual kernel size (0 = identity)
        static void res_unit(mi_unet& u, const std::string& P, int cin, int cout, int rk, int R, bool, int bucket,
                             decltype(add)& add, decltype(wshape)& wshape) {
            for (int su = 0; su < R; ++su) {
                const std::string q = P + ".conv.unit" + std::to_string(su);
                add(q + ".conv.weight", wshape(cout, su == 0 ? cin : cout, 3), bucket);
                add(q + ".conv.bias", {cout}, bucket);
                add(q + ".adn.A.weight", {1}, bucket);
            }
            if (rk) {
                add(P + ".residual.weight", wshape(cout, cin, rk), bucket);
                add(P + ".residual.bias", {cout}, bucket);
            }
        }
    };
    Rec::block(u, "model", 0, add, wshape);
    (void)L; (void)R; (void)c;
    // offsets in backward-completion (bucket) order, 16-byte aligned
    long long off = 0;
    for (int b = 0; b < MI_NUM_BUCKETS; ++b) {
        u.bucket_off[b] = off;
        // within the decoder bucket the top layer finishes first; order by level for buckets 0 and 2
        std::vector<int> idx;
        for (int i = 0; i < (int)u.params.size(); ++i) if (u.params[i].bucket == b) idx.push_back(i);
        auto depth = [&](const std::string& n) { int d = 0; size_t p = 0; while ((p = n.find("submodule", p)) != std::string::npos) { ++d; p += 9; } return d; };
        if (b == 0) std::stable_sort(idx.begin(), idx.end(), [&](int a, int bb) { return depth(u.params[a].name) < depth(u.params[bb].name); });
        if (b == 2) std::stable_sort(idx.begin(), idx.end(), [&](int a, int bb) { return depth(u.params[a].name) > depth(u.params[bb].name); });
        for (int i : idx) { u.params[i].offset = off; off += (u.params[i].numel + 3) / 4 * 4; }
        u.bucket_cnt[b] = off - u.bucket_off[b];
    }
    u.nfloats = off;
}

bool build(mi_unet& u) {
    Builder B(u);
    const int L = u.L, R = u.R, nd = u.nd, dt = u.dt, esz = u.esz, nb = u.nb;
    const int* c = u.cfg.channels;
    const int KK = u.KK;
    const bool train = u.cfg.with_backward != 0;
    int nat[MI_MAX_TAPS];
    for (int i = 0; i < MI_MAX_TAPS; ++i) nat[i] = i;

    // statistics blocks: reserve generously (sum over all norms), exact use tracked by cursors
    {
        size_t tot = 0;
        for (int l = 0; l < L; ++l) tot += (size_t)(3 * R + 4) * nb * MI_NSLOT * (size_t)c[l] * 2 * sizeof(double) + 4096;
        tot += (size_t)MI_NFLAG * 64 + 64;
        u.off_stats = B.alloc(tot); u.stats_bytes = tot;
        u.off_bstats = B.alloc(tot); u.bstats_bytes = tot;
        B.bstats_cursor = align_up(B.bstats_cursor, 64);
        u.off_flags = B.bstats_alloc((size_t)MI_NFLAG * 64);
    }
    std::vector<int> Pcat(L);
    // the skip concatenation [encoder output | decoder output] of level l is kept as two contiguous tensors
    // (A: c[l] channels, B: the rest): consumers take them as a dual-source operand, producers as a dual
    // destination, and every access moves whole cache lines even when a pixel has only 16 channels
    std::vector<Ten> catA(L), catB(L), gcatA(L), gcatB(L);
    for (int l = 0; l <= L - 2; ++l) {
        Pcat[l] = (l < L - 2) ? 2 * c[l] : c[L - 2] + c[L - 1];
        catA[l] = B.new_tensor(l + 1, c[l], esz);
        catB[l] = B.new_tensor(l + 1, Pcat[l] - c[l], esz);
        if (train) { gcatA[l] = B.new_tensor(l + 1, c[l], esz); gcatB[l] = B.new_tensor(l + 1, Pcat[l] - c[l], esz); }
    }
    Ten xin; xin.base = B_XIN; xin.off = 0; xin.pitch = 1; xin.coff = 0;
    Ten logits; logits.base = B_LOGITS; logits.pitch = 1;
    Ten dlogits; dlogits.base = B_DLOGITS; dlogits.pitch = 1;

    struct UnitSave { Ten raw; size_t stats; Ten act; std::string alpha; int lid; size_t wfwd, wdg; std::string wname, bname; };
    struct DownSave { Ten raw0, res; std::vector<UnitSave> un; size_t wf, wd; int ntaps_d; };
    std::vector<DownSave> down(L - 1);
    struct BotSave { std::vector<UnitSave> un; Ten res; bool has_res_conv; size_t wres_f, wres_d; } bot;
    struct UpSave { Ten rawT, actT, rawU; size_t statsT, statsU; int lidT, lidU; size_t wT_f, wT_d, wU_f, wU_d; } ;
    std::vector<UpSave> up(L - 1);

    auto wkey = [&](const std::string& P) { return P + ".conv.weight"; };
    std::vector<std::string> dprefix(L - 1), uprefix(L - 1);
    std::string botprefix;
    {
        std::string P = "model";
        for (int l = 0; l <= L - 2; ++l) {
            dprefix[l] = P + ".0"; uprefix[l] = P + ".2";
            if (l < L - 2) P = P + ".1.submodule"; else botprefix = P + ".1.submodule";
        }
    }
    int kidx[MI_MAX_TAPS], ntaps = 0;

    // ================= forward =================
    B.cur_bucket = 0;
    {
        Step& s = B.push(u.fwd, K_MEMSET, "clear forward statistics");
        REL(s, s.ms, ptr, B_ARENA, u.off_stats); s.ms.bytes = 0;  // bytes patched after the build
        s.kernel = "hipMemsetAsync";
        // Weight packing runs beside the first layers in two launches: the weights of the first encoder levels (few),
        // which the first MFMA convolution waits for, and everything else (the deep levels hold nearly all
        // parameters), which the first deep layer joins.  The split point is patched in after the build.
        const char* pack_name = dt ? "pack_kernel<BF16>" : "pack_kernel<F32>";
        B.push(u.fwd, K_PACK, "pack weights (first levels)").kernel = pack_name;      // push() may move earlier steps
        B.push(u.fwd, K_PACK, "pack weights").kernel = pack_name;
    }
    constexpr int PACK_CUT_LEVEL = 3;              // encoder levels below this one belong to the first launch
    int packs_first = -1; size_t first_rest_step = 0;
    for (int l = 0; l <= L - 2; ++l) {
        const int cin = l == 0 ? 1 : c[l - 1], cout = c[l];
        const std::string P = dprefix[l];
        const std::string lname = "down" + std::to_string(l);
        if (l == PACK_CUT_LEVEL) { packs_first = (int)u.packs.size(); first_rest_step = u.fwd.size(); }
        DownSave& D = down[l];
        D.raw0 = B.new_tensor(l + 1, cout, esz);   // unit0 conv output (before the norm)
        D.res = B.new_tensor(l + 1, cout, esz);    // residual conv output
        D.un.resize(R);
        for (int su = 0; su < R; ++su) {
            D.un[su].stats = B.stats_alloc((size_t)nb * MI_NSLOT * cout * 2 * sizeof(double));
            D.un[su].alpha = P + ".conv.unit" + std::to_string(su) + ".adn.A.weight";
            D.un[su].lid = B.layer_id++;
            D.un[su].wname = P + ".conv.unit" + std::to_string(su) + ".conv.weight";
            D.un[su].bname = P + ".conv.unit" + std::to_string(su) + ".conv.bias";
        }
        if (l == 0) {
            Step& s = B.push(u.fwd, K_C1_FWD, lname + ".unit0+residual conv s2 (1->2C)");
            C1Params& p = s.c1;
            REL(s, p, x1, B_XIN, 0);
            REL(s, p, y_out, B_ARENA, D.raw0.off); p.y_pitch = cout; p.y_coff = 0; p.N = 2 * cout;
            REL(s, p, y2_out, B_ARENA, D.res.off); p.y2o_pitch = cout; p.y2o_coff = 0; p.n_o1 = cout;
            REL(s, p, w, B_PARAMS, B.poff(D.un[0].wname) * sizeof(float));
            REL(s, p, w2, B_PARAMS, B.poff(P + ".residual.weight") * sizeof(float)); p.n_w1 = cout;
            REL(s, p, bias, B_PARAMS, B.poff(P + ".residual.bias") * sizeof(float)); p.bias_c0 = cout;
            REL(s, p, stats, B_ARENA, D.un[0].stats); p.stats_c = cout; p.nstat = cout;
            p.ndim = nd; p.nb = nb;
            for (int d = 0; d < 3; ++d) { p.fd[d] = u.rd[0][d]; p.cd[d] = u.rd[1][d]; }
            s.kernel = dt ? "c1_s2_fwd_kernel<BF16>" : "c1_s2_fwd_kernel<F32>";
            s.work = (double)nb * (u.rp[0] * 4.0 + u.rp[1] * 2.0 * cout * esz);
            u.flops_fwd += 2.0 * nb * u.rp[1] * KK * 2.0 * cout;
        } else {
            ConvShape cs = B.shape(0, 3, 2, cin, 2 * cout, l);
            D.wf = B.packed_alloc(KK, 2 * cout, cin);
            Ten in = catA[l - 1];
            if (!B.add_igemm(u.fwd, lname + ".unit0+residual conv s2", cs, DIR_FWD, in, cin, nullptr, cin, 2 * cout, D.wf,
                             (P + ".residual.bias").c_str(), cout, D.raw0, nullptr, D.un[0].stats, cout, cout, kidx, &ntaps,
                             &D.res, cout))
                return false;
            B.add_pack(false, D.wf, D.un[0].wname, cout, cin, ntaps, kidx, cin * KK, KK, 2 * cout, cin, 0, 0);
            B.add_pack(false, D.wf, P + ".residual.weight", cout, cin, ntaps, kidx, cin * KK, KK, 2 * cout, cin, cout, 0);
        }
        D.un[0].raw = D.raw0;
        for (int su = 0; su < R; ++su) {
            Ten res = D.res;
            if (su < R - 1) {
                D.un[su].act = B.new_tensor(l + 1, cout, esz);
                B.add_norm_fwd(lname + ".unit" + std::to_string(su) + " norm+prelu", l + 1, cout, dt, D.un[su].raw,
                               D.un[su].stats, D.un[su].alpha, nullptr, D.un[su].act, D.un[su].lid);
                D.un[su + 1].raw = B.new_tensor(l + 1, cout, esz);
                ConvShape cs = B.shape(0, 3, 1, cout, cout, l + 1);
                D.un[su + 1].wfwd = B.packed_alloc(KK, cout, cout);
                if (!B.add_igemm(u.fwd, lname + ".unit" + std::to_string(su + 1) + " conv", cs, DIR_FWD, D.un[su].act, cout,
                                 nullptr, cout, cout, D.un[su + 1].wfwd, nullptr, 0, D.un[su + 1].raw, nullptr,
                                 D.un[su + 1].stats, cout, cout, kidx, &ntaps))
                    return false;
                B.add_pack(false, D.un[su + 1].wfwd, D.un[su + 1].wname, cout, cout, ntaps, kidx, cout * KK, KK, cout, cout, 0, 0);
            } else {
                Ten out = catA[l];
                B.add_norm_fwd(lname + ".unit" + std::to_string(su) + " norm+prelu+residual", l + 1, cout, dt, D.un[su].raw,
                               D.un[su].stats, D.un[su].alpha, &res, out, D.un[su].lid);
            }
        }
    }
    // ---- bottom ----
    if (packs_first < 0) { packs_first = (int)u.packs.size(); first_rest_step = u.fwd.size(); }
    {
        const int cin = c[L - 2], cout = c[L - 1], res = L - 1;
        const std::string P = botprefix;
        bot.un.resize(R);
        bot.has_res_conv = cin != cout;
        Ten in = catA[L - 2];
        for (int su = 0; su < R; ++su) {
            UnitSave& U = bot.un[su];
            U.stats = B.stats_alloc((size_t)nb * MI_NSLOT * cout * 2 * sizeof(double));
            U.alpha = P + ".conv.unit" + std::to_string(su) + ".adn.A.weight";
            U.wname = P + ".conv.unit" + std::to_string(su) + ".conv.weight";
            U.bname = P + ".conv.unit" + std::to_string(su) + ".conv.bias";
            U.lid = B.layer_id++;
            U.raw = B.new_tensor(res, cout, esz);
            const int ci = su == 0 ? cin : cout;
            ConvShape cs = B.shape(0, 3, 1, ci, cout, res);
            U.wfwd = B.packed_alloc(KK, cout, ci);
            const Ten& src = su == 0 ? in : bot.un[su - 1].act;
            if (!B.add_igemm(u.fwd, "bottom.unit" + std::to_string(su) + " conv", cs, DIR_FWD, src, ci, nullptr, ci, cout,
                             U.wfwd, nullptr, 0, U.raw, nullptr, U.stats, cout, cout, kidx, &ntaps))
                return false;
            B.add_pack(false, U.wfwd, U.wname, cout, ci, ntaps, kidx, ci * KK, KK, cout, ci, 0, 0);
            if (su == 0) {
                if (bot.has_res_conv) {
                    bot.res = B.new_tensor(res, cout, esz);
                    ConvShape cr = B.shape(0, 1, 1, cin, cout, res);
                    bot.wres_f = B.packed_alloc(1, cout, cin);
                    if (!B.add_igemm(u.fwd, "bottom.residual conv 1x1", cr, DIR_FWD, in, cin, nullptr, cin, cout, bot.wres_f,
                                     (P + ".residual.bias").c_str(), 0, bot.res, nullptr, (size_t)-1, 0, 0, kidx, &ntaps))
                        return false;
                    B.add_pack(false, bot.wres_f, P + ".residual.weight", cout, cin, ntaps, kidx, cin, 1, cout, cin, 0, 0);
                } else {
                    bot.res = in;
                }
            }
            if (su < R - 1) {
                U.act = B.new_tensor(res, cout, esz);
                B.add_norm_fwd("bottom.unit" + std::to_string(su) + " norm+prelu", res, cout, dt, U.raw, U.stats, U.alpha,
                               nullptr, U.act, U.lid);
            } else {
                Ten out = catB[L - 2];
                B.add_norm_fwd("bottom.unit" + std::to_string(su) + " norm+prelu+residual", res, cout, dt, U.raw, U.stats,
                               U.alpha, &bot.res, out, U.lid);
            }
        }
    }
    // ---- up path ----
    for (int l = L - 2; l >= 0; --l) {
        const std::string P = uprefix[l];
        const std::string lname = "up" + std::to_string(l);
        UpSave& U = up[l];
        const int K = Pcat[l];
        const int outc = l > 0 ? c[l - 1] : 1;
        U.lidT = B.layer_id++; U.lidU = B.layer_id++;
        if (l > 0) {
            U.rawT = B.new_tensor(l, outc, esz); U.actT = B.new_tensor(l, outc, esz); U.rawU = B.new_tensor(l, outc, esz);
            U.statsT = B.stats_alloc((size_t)nb * MI_NSLOT * outc * 2 * sizeof(double));
            U.statsU = B.stats_alloc((size_t)nb * MI_NSLOT * outc * 2 * sizeof(double));
            ConvShape ct = B.shape(1, 3, 2, K, outc, l + 1);
            U.wT_f = B.packed_alloc(KK, outc, K);
            if (!B.add_igemm(u.fwd, lname + ".convT", ct, DIR_FWD, catA[l], c[l], &catB[l], K, outc, U.wT_f, nullptr, 0, U.rawT,
                             nullptr, U.statsT, outc, outc, kidx, &ntaps))
                return false;
            B.add_pack(false, U.wT_f, P + ".0.conv.weight", outc, K, ntaps, kidx, KK, outc * KK, outc, K, 0, 0);
            B.add_norm_fwd(lname + ".convT norm+prelu", l, outc, dt, U.rawT, U.statsT, P + ".0.adn.A.weight", nullptr, U.actT, U.lidT);
            ConvShape cu = B.shape(0, 3, 1, outc, outc, l);
            U.wU_f = B.packed_alloc(KK, outc, outc);
            if (!B.add_igemm(u.fwd, lname + ".ru conv", cu, DIR_FWD, U.actT, outc, nullptr, outc, outc, U.wU_f, nullptr, 0,
                             U.rawU, nullptr, U.statsU, outc, outc, kidx, &ntaps))
                return false;
            B.add_pack(false, U.wU_f, P + ".1.conv.unit0.conv.weight", outc, outc, ntaps, kidx, outc * KK, KK, outc, outc, 0, 0);
            Ten out = catB[l - 1];
            B.add_norm_fwd(lname + ".ru norm+prelu+residual", l, outc, dt, U.rawU, U.statsU, P + ".1.conv.unit0.adn.A.weight",
                           &U.actT, out, U.lidU);
        } else {
            U.rawT = B.new_tensor(0, 1, 4); U.actT = B.new_tensor(0, 1, 4);
            U.statsT = B.stats_alloc((size_t)nb * MI_NSLOT * 2 * sizeof(double));
            {
                Step& s = B.push(u.fwd, K_C1_T, lname + ".convT (2C->1)");
                C1Params& p = s.c1;
                REL(s, p, y, B_ARENA, catA[0].off); p.y_pitch = c[0]; p.y_coff = 0; p.N = K;
                REL(s, p, y2, B_ARENA, catB[0].off); p.y2_pitch = K - c[0]; p.y2_coff = 0; p.n_y1 = c[0];
                REL(s, p, x1_out, B_ARENA, U.rawT.off);
                REL(s, p, w, B_PARAMS, B.poff(P + ".0.conv.weight") * sizeof(float));
                REL(s, p, stats, B_ARENA, U.statsT); p.stats_c = 1; p.nstat = 1;
                p.ndim = nd; p.nb = nb;
                for (int d = 0; d < 3; ++d) { p.fd[d] = u.rd[0][d]; p.cd[d] = u.rd[1][d]; }
                s.kernel = dt ? "c1_s2_T_kernel<BF16>" : "c1_s2_T_kernel<F32>";
                s.work = (double)nb * (u.rp[0] * 4.0 + u.rp[1] * (double)K * esz);
                u.flops_fwd += 2.0 * nb * u.rp[1] * KK * (double)K;
            }
            // The one-channel top: when the 2-D strip stencil serves the shape it reads the raw transposed-convolution
            // output and normalises, drops and activates it on load (MI_NO_NORM_ON_LOAD=1: the separate pass); the
            // activated tensor is written on the side when a backward pass will read it.
            StencilParams probe; memset(&probe, 0, sizeof(probe));
            probe.ndim = nd; for (int d = 0; d < 3; ++d) probe.d[d] = u.rd[0][d];
            static const bool no_fuse = getenv("MI_NO_NORM_ON_LOAD") && atoi(getenv("MI_NO_NORM_ON_LOAD")) != 0;
            const bool fuse_top = !no_fuse && stencil_norm_on_load_ok(probe);
            if (fuse_top) B.record_norm(0, 1, 0, U.rawT, U.statsT, U.lidT);
            else B.add_norm_fwd(lname + ".convT norm+prelu", 0, 1, 0, U.rawT, U.statsT, P + ".0.adn.A.weight", nullptr, U.actT, U.lidT);
            {
                Step& s = B.push(u.fwd, K_STENCIL, lname + (fuse_top ? ".convT norm+prelu on load, ru conv (1->1) + identity" : ".ru conv (1->1) + identity"));
                StencilParams& p = s.stn;
                REL(s, p, in, B_ARENA, fuse_top ? U.rawT.off : U.actT.off);
                REL(s, p, w, B_PARAMS, B.poff(P + ".1.conv.unit0.conv.weight") * sizeof(float));
                REL(s, p, bias, B_PARAMS, B.poff(P + ".1.conv.unit0.conv.bias") * sizeof(float));
                if (fuse_top) {
                    REL(s, p, nrm_stats, B_ARENA, U.statsT);
                    REL(s, p, nrm_alpha, B_PARAMS, B.poff(P + ".0.adn.A.weight") * sizeof(float));
                    p.nrm_eps = 1e-5f; p.nrm_layer = (unsigned)U.lidT;
                    if (train) REL(s, p, act_out, B_ARENA, U.actT.off);
                } else {
                    REL(s, p, addend, B_ARENA, U.actT.off);
                }
                REL(s, p, out, B_LOGITS, 0);
                p.flip = 0; p.ndim = nd; p.nb = nb;
                for (int d = 0; d < 3; ++d) p.d[d] = u.rd[0][d];
                s.kernel = "stencil_kernel";
                s.work = (double)nb * u.rp[0] * 4.0 * (fuse_top && train ? 3.0 : 2.0);
                u.flops_fwd += 2.0 * nb * u.rp[0] * KK;
            }
        }
    }
    const int n_fwd_packs = (int)u.packs.size();
    u.n_fwd_packs = n_fwd_packs;

    // ================= backward =================
    if (train) {
        B.cur_bucket = 0;
        {
            Step& s = B.push(u.bwd, K_MEMSET, "clear backward statistics");
            REL(s, s.ms, ptr, B_ARENA, u.off_bstats); s.ms.bytes = 0;
            s.kernel = "hipMemsetAsync";
        }
        size_t po, pb; int nslab, nt;
        // ---- top (up0) ----
        {
            const std::string P = uprefix[0];
            UpSave& U = up[0];
            const int K = Pcat[0];
            Ten g_actT = B.new_tensor(0, 1, 4), g_rawT = B.new_tensor(0, 1, 4);
            {
                Step& s = B.push(u.bwd, K_STENCIL, "up0.ru conv dgrad + identity");
                StencilParams& p = s.stn;
                REL(s, p, in, B_DLOGITS, 0);
                REL(s, p, w, B_PARAMS, B.poff(P + ".1.conv.unit0.conv.weight") * sizeof(float));
                REL(s, p, addend, B_DLOGITS, 0);
                REL(s, p, out, B_ARENA, g_actT.off);
                p.flip = 1; p.ndim = nd; p.nb = nb;
                for (int d = 0; d < 3; ++d) p.d[d] = u.rd[0][d];
                s.kernel = "stencil_kernel";
                s.work = (double)nb * u.rp[0] * 4.0 * 2.0;
                u.flops_bwd += 2.0 * nb * u.rp[0] * KK;
            }
            {
                Step& s = B.push(u.bwd, K_STENCIL_WGRAD, "up0.ru conv wgrad");
                StencilParams& p = s.stn;
                REL(s, p, in, B_ARENA, U.actT.off);
                REL(s, p, dout, B_DLOGITS, 0);
                p.ndim = nd; p.nb = nb;
                for (int d = 0; d < 3; ++d) p.d[d] = u.rd[0][d];
                {
                    const long long items = (long long)nb * u.rp[0] / (p.d[2] % 4 == 0 ? 4 : 1);
                    long long blocks = (items + 255) / 256;
                    p.nsplit = (int)(blocks < 1 ? 1 : blocks > 1024 ? 1024 : blocks);
                }
                po = B.alloc((size_t)p.nsplit * KK * sizeof(float)); pb = B.alloc((size_t)p.nsplit * sizeof(float));
                REL(s, p, partial, B_ARENA, po); REL(s, p, partial_bias, B_ARENA, pb);
                s.kernel = "stencil_wgrad_kernel";
                s.work = (double)nb * u.rp[0] * 4.0 * 2.0;
                u.flops_bwd += 2.0 * nb * u.rp[0] * KK;
                B.add_reduce(po, pb, p.nsplit, KK, 1, 1, 0, 1, KK, nat, P + ".1.conv.unit0.conv.weight", P + ".1.conv.unit0.conv.bias");
            }
            B.add_norm_bwd("up0.convT norm+prelu bwd", 0, 1, 0, g_actT, U.rawT, U.statsT, P + ".0.adn.A.weight", g_rawT, U.lidT);
            {
                Step& s = B.push(u.bwd, K_C1_FWD, "up0.convT dgrad (1->2C conv s2)");
                C1Params& p = s.c1;
                REL(s, p, x1, B_ARENA, g_rawT.off);
                REL(s, p, y_out, B_ARENA, gcatA[0].off); p.y_pitch = c[0]; p.y_coff = 0; p.N = K;
                REL(s, p, y2_out, B_ARENA, gcatB[0].off); p.y2o_pitch = K - c[0]; p.y2o_coff = 0; p.n_o1 = c[0];
                REL(s, p, w, B_PARAMS, B.poff(P + ".0.conv.weight") * sizeof(float)); p.n_w1 = K;
                p.ndim = nd; p.nb = nb;
                for (int d = 0; d < 3; ++d) { p.fd[d] = u.rd[0][d]; p.cd[d] = u.rd[1][d]; }
                s.kernel = dt ? "c1_s2_fwd_kernel<BF16>" : "c1_s2_fwd_kernel<F32>";
                s.work = (double)nb * (u.rp[0] * 4.0 + u.rp[1] * (double)K * esz);
                u.flops_bwd += 2.0 * nb * u.rp[1] * KK * (double)K;
            }
            {
                Step& s = B.push(u.bwd, K_C1_WGRAD, "up0.convT wgrad");
                C1Params& p = s.c1;
                REL(s, p, y, B_ARENA, catA[0].off); p.y_pitch = c[0]; p.y_coff = 0; p.N = K;
                REL(s, p, y2, B_ARENA, catB[0].off); p.y2_pitch = K - c[0]; p.y2_coff = 0;
                REL(s, p, x1, B_ARENA, g_rawT.off);
                p.ndim = nd; p.nb = nb; p.n_y1 = c[0];
                for (int d = 0; d < 3; ++d) { p.fd[d] = u.rd[0][d]; p.cd[d] = u.rd[1][d]; }
                p.nsplit = c1_wgrad_slabs(p.N, p.cd);
                po = B.alloc((size_t)p.nsplit * KK * K * sizeof(float)); pb = B.alloc((size_t)p.nsplit * K * sizeof(float));
                REL(s, p, partial, B_ARENA, po); REL(s, p, partial_bias, B_ARENA, pb);
                s.kernel = c1_wgrad3d_mfma_applicable(dt, p) ? "c1_s2_wgrad3d_mfma_kernel" : dt ? "c1_s2_wgrad_kernel<BF16>" : "c1_s2_wgrad_kernel<F32>";
                s.work = (double)nb * (u.rp[0] * 4.0 + u.rp[1] * (double)K * esz);
                u.flops_bwd += 2.0 * nb * u.rp[1] * KK * (double)K;
                B.add_reduce(po, pb, p.nsplit, KK, K, 1, 0, K, KK, nat, P + ".0.conv.weight", "");
                B.add_scalar_reduce(0, 0, P + ".0.conv.bias");  // feeds an instance norm: gradient is exactly zero
            }
        }
        // ---- up l = 1 .. L-2 ----
        for (int l = 1; l <= L - 2; ++l) {
            const std::string P = uprefix[l];
            const std::string lname = "up" + std::to_string(l);
            UpSave& U = up[l];
            const int outc = c[l - 1], K = Pcat[l];
            Ten dout = gcatB[l - 1];
            Ten g_rawU = B.new_tensor(l, outc, esz), g_actT = B.new_tensor(l, outc, esz), g_rawT = B.new_tensor(l, outc, esz);
            B.add_norm_bwd(lname + ".ru norm+prelu bwd", l, outc, dt, dout, U.rawU, U.statsU, P + ".1.conv.unit0.adn.A.weight", g_rawU, U.lidU);
            ConvShape cu = B.shape(0, 3, 1, outc, outc, l);
            U.wU_d = B.packed_alloc(KK, outc, outc);
            if (!B.add_igemm(u.bwd, lname + ".ru conv dgrad", cu, DIR_DGRAD, g_rawU, outc, nullptr, outc, outc, U.wU_d, nullptr, 0,
                             g_actT, &dout, (size_t)-1, 0, 0, kidx, &ntaps))
                return false;
            B.add_pack(true, U.wU_d, P + ".1.conv.unit0.conv.weight", outc, outc, ntaps, kidx, KK, outc * KK, outc, outc, 0, 0);
            B.add_wgrad(lname + ".ru conv wgrad", cu, g_rawU, outc, nullptr, outc, U.actT, outc, &po, &pb, &nslab, &nt);
            B.add_reduce(po, pb, nslab, nt, outc, outc, 0, outc, KK, nat, P + ".1.conv.unit0.conv.weight", P + ".1.conv.unit0.conv.bias");
            B.add_norm_bwd(lname + ".convT norm+prelu bwd", l, outc, dt, g_actT, U.rawT, U.statsT, P + ".0.adn.A.weight", g_rawT, U.lidT);
            ConvShape ct = B.shape(1, 3, 2, K, outc, l + 1);
            U.wT_d = B.packed_alloc(KK, K, outc);
            if (!B.add_igemm(u.bwd, lname + ".convT dgrad", ct, DIR_DGRAD, g_rawT, outc, nullptr, outc, K, U.wT_d, nullptr, 0,
                             gcatA[l], nullptr, (size_t)-1, 0, 0, kidx, &ntaps, &gcatB[l], c[l]))
                return false;
            B.add_pack(true, U.wT_d, P + ".0.conv.weight", K, outc, ntaps, kidx, outc * KK, KK, K, outc, 0, 0);
            B.add_wgrad(lname + ".convT wgrad", ct, catA[l], c[l], &catB[l], K, g_rawT, outc, &po, &pb, &nslab, &nt);
            B.add_reduce(po, pb, nslab, nt, K, outc, 0, K, KK, nat, P + ".0.conv.weight", "");
            B.add_scalar_reduce(0, 0, P + ".0.conv.bias", outc);
        }
        B.push_reduce_step("reduce gradients: decoder bucket");
        // ---- bottom ----
        B.cur_bucket = 1;
        {
            const int cin = c[L - 2], cout = c[L - 1], res = L - 1;
            const std::string P = botprefix;
            Ten dout = gcatB[L - 2];
            Ten gin = gcatA[L - 2];
            Ten in = catA[L - 2];
            Ten g_raw = B.new_tensor(res, cout, esz);
            B.add_norm_bwd("bottom.unit" + std::to_string(R - 1) + " norm+prelu bwd", res, cout, dt, dout, bot.un[R - 1].raw,
                           bot.un[R - 1].stats, bot.un[R - 1].alpha, g_raw, bot.un[R - 1].lid);
            for (int su = R - 1; su >= 1; --su) {
                UnitSave& U = bot.un[su];
                Ten g_act = B.new_tensor(res, cout, esz);
                ConvShape cs = B.shape(0, 3, 1, cout, cout, res);
                U.wdg = B.packed_alloc(KK, cout, cout);
                if (!B.add_igemm(u.bwd, "bottom.unit" + std::to_string(su) + " conv dgrad", cs, DIR_DGRAD, g_raw, cout, nullptr,
                                 cout, cout, U.wdg, nullptr, 0, g_act, nullptr, (size_t)-1, 0, 0, kidx, &ntaps))
                    return false;
                B.add_pack(true, U.wdg, U.wname, cout, cout, ntaps, kidx, KK, cout * KK, cout, cout, 0, 0);
                B.add_wgrad("bottom.unit" + std::to_string(su) + " conv wgrad", cs, g_raw, cout, nullptr, cout, bot.un[su - 1].act,
                            cout, &po, &pb, &nslab, &nt);
                B.add_reduce(po, pb, nslab, nt, cout, cout, 0, cout, KK, nat, U.wname, U.bname);
                Ten g_raw_prev = B.new_tensor(res, cout, esz);
                B.add_norm_bwd("bottom.unit" + std::to_string(su - 1) + " norm+prelu bwd", res, cout, dt, g_act, bot.un[su - 1].raw,
                               bot.un[su - 1].stats, bot.un[su - 1].alpha, g_raw_prev, bot.un[su - 1].lid);
                g_raw = g_raw_prev;
            }
            {
                UnitSave& U = bot.un[0];
                ConvShape cs = B.shape(0, 3, 1, cin, cout, res);
                U.wdg = B.packed_alloc(KK, cin, cout);
                if (!B.add_igemm(u.bwd, "bottom.unit0 conv dgrad (+=)", cs, DIR_DGRAD, g_raw, cout, nullptr, cout, cin, U.wdg,
                                 nullptr, 0, gin, &gin, (size_t)-1, 0, 0, kidx, &ntaps))
                    return false;
                B.add_pack(true, U.wdg, U.wname, cin, cout, ntaps, kidx, KK, cin * KK, cin, cout, 0, 0);
                B.add_wgrad("bottom.unit0 conv wgrad", cs, g_raw, cout, nullptr, cout, in, cin, &po, &pb, &nslab, &nt);
                B.add_reduce(po, pb, nslab, nt, cout, cin, 0, cout, KK, nat, U.wname, U.bname);
            }
            if (bot.has_res_conv) {
                ConvShape cr = B.shape(0, 1, 1, cin, cout, res);
                bot.wres_d = B.packed_alloc(1, cin, cout);
                if (!B.add_igemm(u.bwd, "bottom.residual conv 1x1 dgrad (+=)", cr, DIR_DGRAD, dout, cout, nullptr, cout, cin,
                                 bot.wres_d, nullptr, 0, gin, &gin, (size_t)-1, 0, 0, kidx, &ntaps))
                    return false;
                B.add_pack(true, bot.wres_d, P + ".residual.weight", cin, cout, ntaps, kidx, 1, cin, cin, cout, 0, 0);
                B.add_wgrad("bottom.residual conv 1x1 wgrad", cr, dout, cout, nullptr, cout, in, cin, &po, &pb, &nslab, &nt);
                B.add_reduce(po, pb, nslab, nt, cout, cin, 0, cout, 1, nat, P + ".residual.weight", P + ".residual.bias");
            } else {
                Step& s = B.push(u.bwd, K_ADD, "bottom identity residual (+=)");
                REL(s, s.add, dst, B_ARENA, gin.off); s.add.dpitch = gin.pitch; s.add.dcoff = 0;
                REL(s, s.add, src, B_ARENA, dout.off); s.add.spitch = dout.pitch; s.add.scoff = dout.coff;
                s.add.c = cin; s.add.npix = (long long)nb * u.rp[res];
                s.kernel = "add_kernel";
                s.work = (double)nb * u.rp[res] * cin * esz * 3.0;
            }
        }
        B.push_reduce_step("reduce gradients: bottom bucket");
        // ---- down path ----
        B.cur_bucket = 2;
        for (int l = L - 2; l >= 0; --l) {
            const int cin = l == 0 ? 1 : c[l - 1], cout = c[l];
            const std::string P = dprefix[l];
            const std::string lname = "down" + std::to_string(l);
            DownSave& D = down[l];
            // the last weight gradients run largely after the main chain has ended, with less to slow down: one workgroup per CU
            B.wgrad_target = l == 0 ? (getenv("MI_WG_TAIL") ? atoi(getenv("MI_WG_TAIL")) : 256) : 0;
            Ten dout = gcatA[l];
            Ten g_raw = B.new_tensor(l + 1, cout, esz);
            B.add_norm_bwd(lname + ".unit" + std::to_string(R - 1) + " norm+prelu bwd", l + 1, cout, dt, dout, D.un[R - 1].raw,
                           D.un[R - 1].stats, D.un[R - 1].alpha, g_raw, D.un[R - 1].lid);
            for (int su = R - 1; su >= 1; --su) {
                UnitSave& U = D.un[su];
                Ten g_act = B.new_tensor(l + 1, cout, esz);
                ConvShape cs = B.shape(0, 3, 1, cout, cout, l + 1);
                U.wdg = B.packed_alloc(KK, cout, cout);
                if (!B.add_igemm(u.bwd, lname + ".unit" + std::to_string(su) + " conv dgrad", cs, DIR_DGRAD, g_raw, cout, nullptr,
                                 cout, cout, U.wdg, nullptr, 0, g_act, nullptr, (size_t)-1, 0, 0, kidx, &ntaps))
                    return false;
                B.add_pack(true, U.wdg, U.wname, cout, cout, ntaps, kidx, KK, cout * KK, cout, cout, 0, 0);
                B.add_wgrad(lname + ".unit" + std::to_string(su) + " conv wgrad", cs, g_raw, cout, nullptr, cout, D.un[su - 1].act,
                            cout, &po, &pb, &nslab, &nt);
                B.add_reduce(po, pb, nslab, nt, cout, cout, 0, cout, KK, nat, U.wname, U.bname);
                Ten g_raw_prev = B.new_tensor(l + 1, cout, esz);
                B.add_norm_bwd(lname + ".unit" + std::to_string(su - 1) + " norm+prelu bwd", l + 1, cout, dt, g_act, D.un[su - 1].raw,
                               D.un[su - 1].stats, D.un[su - 1].alpha, g_raw_prev, D.un[su - 1].lid);
                g_raw = g_raw_prev;
            }
            if (l > 0) {
                ConvShape cs = B.shape(0, 3, 2, cin, 2 * cout, l);
                Ten gin = gcatA[l - 1];
                Ten in = catA[l - 1];
                D.wd = B.packed_alloc(KK, cin, 2 * cout);
                if (!B.add_igemm(u.bwd, lname + ".unit0+residual conv s2 dgrad (+=)", cs, DIR_DGRAD, g_raw, cout, &dout, 2 * cout,
                                 cin, D.wd, nullptr, 0, gin, &gin, (size_t)-1, 0, 0, kidx, &ntaps))
                    return false;
                B.add_pack(true, D.wd, D.un[0].wname, cin, cout, ntaps, kidx, KK, cin * KK, cin, 2 * cout, 0, 0);
                B.add_pack(true, D.wd, P + ".residual.weight", cin, cout, ntaps, kidx, KK, cin * KK, cin, 2 * cout, 0, cout);
                B.add_wgrad(lname + ".unit0+residual conv s2 wgrad", cs, g_raw, cout, &dout, 2 * cout, in, cin, &po, &pb, &nslab, &nt);
                B.add_reduce(po, pb, nslab, nt, 2 * cout, cin, 0, cout, KK, nat, D.un[0].wname, D.un[0].bname);
                B.add_reduce(po, pb, nslab, nt, 2 * cout, cin, cout, cout, KK, nat, P + ".residual.weight", P + ".residual.bias");
            } else {
                {
                    // unit0 (gradient of the raw conv output) and the residual conv (gradient of the block output)
                    // share the input image: one launch, two gradient sources
                    Step& s = B.push(u.bwd, K_C1_WGRAD, "down0.unit0+residual conv wgrad");
                    C1Params& p = s.c1;
                    REL(s, p, y, B_ARENA, g_raw.off); p.y_pitch = g_raw.pitch; p.y_coff = g_raw.coff;
                    REL(s, p, y2, B_ARENA, dout.off); p.y2_pitch = dout.pitch; p.y2_coff = dout.coff;
                    p.N = 2 * cout; p.n_y1 = cout;
                    REL(s, p, x1, B_XIN, 0);
                    p.ndim = nd; p.nb = nb;
                    for (int d = 0; d < 3; ++d) { p.fd[d] = u.rd[0][d]; p.cd[d] = u.rd[1][d]; }
                    p.nsplit = c1_wgrad_slabs(p.N, p.cd);
                    po = B.alloc((size_t)p.nsplit * KK * 2 * cout * sizeof(float)); pb = B.alloc((size_t)p.nsplit * 2 * cout * sizeof(float));
                    REL(s, p, partial, B_ARENA, po); REL(s, p, partial_bias, B_ARENA, pb);
                    s.kernel = c1_wgrad3d_mfma_applicable(dt, p) ? "c1_s2_wgrad3d_mfma_kernel" : dt ? "c1_s2_wgrad_kernel<BF16>" : "c1_s2_wgrad_kernel<F32>";
                    s.work = (double)nb * (u.rp[0] * 4.0 + u.rp[1] * 2.0 * cout * esz);
                    u.flops_bwd += 2.0 * nb * u.rp[1] * KK * 2.0 * cout;
                    B.add_reduce(po, pb, p.nsplit, KK, 2 * cout, 1, 0, cout, KK, nat, D.un[0].wname, D.un[0].bname);
                    B.add_reduce(po, pb, p.nsplit, KK, 2 * cout, 1, cout, cout, KK, nat, P + ".residual.weight", P + ".residual.bias");
                }
            }
            // everything but level 0 is reduced while level 0 is still being differentiated, so that the reduction left
            // behind the last weight gradient (nothing else runs then) covers level 0 only; level 0 is a bucket of its
            // own, so a data-parallel caller's all-reduce of the deeper encoder levels runs beside level 0's backward
            if (l == 1) { B.push_reduce_step("reduce gradients: encoder bucket, levels 1 and deeper"); B.cur_bucket = 3; }
        }
        B.cur_bucket = 3;
        B.push_reduce_step("reduce gradients: encoder bucket, level 0");
    }

    // ---- descriptor tables: the ranges of descriptors each table step launches ----
    // reduces are grouped by their K_REDUCE step in creation order
    {
        std::vector<int> first(B.n_reduce_steps, -1), cnt(B.n_reduce_steps, 0);
        for (size_t i = 0; i < B.group_of_reduce.size(); ++i) {
            const int g = B.group_of_reduce[i];
            if (g >= B.n_reduce_steps) { mi_set_error("gradient reduction without a launch"); return false; }
            if (first[g] < 0) first[g] = (int)i;
            cnt[g]++;
        }
        int k = 0;
        for (Step& s : u.bwd)
            if (s.kind == K_REDUCE) {
                s.tab.first = first[k] < 0 ? 0 : first[k]; s.tab.count = cnt[k];
                for (int i = s.tab.first; i < s.tab.first + s.tab.count; ++i)
                    u.reduces[i].nblk = reduce_blocks_for(u.reduces[i].ntaps * u.reduces[i].Fcnt * u.reduces[i].S + u.reduces[i].Fcnt);
                ++k;
            }
    }
    {
        // first packing launch: descriptors [0, packs_first); second: the rest (forward and backward packs in one launch)
        const int total = train ? (int)u.packs.size() : n_fwd_packs;
        int npack = 0;
        bool first_igemm = true;
        for (size_t i = 0; i < u.fwd.size(); ++i) {
            Step& s = u.fwd[i];
            if (s.kind == K_PACK) {
                if (npack++ == 0) { s.tab.first = 0; s.tab.count = packs_first; }
                else { s.tab.first = packs_first; s.tab.count = total - packs_first; }
                for (int j = s.tab.first; j < s.tab.first + s.tab.count; ++j) u.packs[j].nblk = pack_blocks_for(u.packs[j].N * u.packs[j].K);
            }
            if (s.kind == K_IGEMM && first_igemm) { s.wait_pack = 1; first_igemm = false; }
            if (i == first_rest_step) s.join_side = 1;
            if (s.kind == K_MEMSET) s.ms.bytes = align_up(B.stats_cursor, 16);
        }
        if (first_rest_step >= u.fwd.size() || u.fwd[first_rest_step].kind != K_IGEMM) { mi_set_error("packing split: no convolution at the split point"); return false; }
    }
    for (Step& s : u.bwd)
        if (s.kind == K_MEMSET) s.ms.bytes = align_up(B.bstats_cursor, 16);
    if (B.stats_cursor > u.stats_bytes || B.bstats_cursor > u.bstats_bytes) { mi_set_error("statistics region overflow"); return false; }
    return true;
}

// Table steps: the descriptors of the step, pointers resolved against the caller's buffers, go out as kernel
// arguments in chunks of one table per launch.
int run_pack(mi_unet& u, const Step& s, char* const* bases, hipStream_t st) {
    for (int i0 = s.tab.first; i0 < s.tab.first + s.tab.count; i0 += MI_PACK_TAB) {
        PackTable t; memset(&t, 0, sizeof(t));
        int blk = 0;
        for (int i = i0; i < s.tab.first + s.tab.count && t.n < MI_PACK_TAB; ++i) {
            PackDesc& d = t.d[t.n++];
            d = u.packs[i];
            d.src = reinterpret_cast<const float*>(bases[B_PARAMS] + reinterpret_cast<uintptr_t>(d.src));
            d.dst = bases[B_ARENA] + reinterpret_cast<uintptr_t>(d.dst);
            d.blk0 = blk; blk += d.nblk;
        }
        if (int r = launch_pack(s.dtype, t, blk, st)) return r;
    }
    return 0;
}

int run_reduce(mi_unet& u, const Step& s, char* const* bases, hipStream_t st) {
    if (!bases[B_GRADS]) { mi_set_error("gradient reduction needs the gradient buffer"); return -1; }
    for (int i0 = s.tab.first; i0 < s.tab.first + s.tab.count; i0 += MI_REDUCE_TAB) {
        ReduceTable t; memset(&t, 0, sizeof(t));
        int blk = 0;
        for (int i = i0; i < s.tab.first + s.tab.count && t.n < MI_REDUCE_TAB; ++i) {
            ReduceDesc& d = t.d[t.n++];
            d = u.reduces[i];
            d.partial = reinterpret_cast<const float*>(bases[B_ARENA] + reinterpret_cast<uintptr_t>(d.partial));
            d.partial_bias = reinterpret_cast<const float*>(bases[B_ARENA] + reinterpret_cast<uintptr_t>(d.partial_bias));
            d.dst_w = reinterpret_cast<float*>(bases[B_GRADS] + reinterpret_cast<uintptr_t>(d.dst_w));
            d.dst_b = u.reduce_has_b[i] ? reinterpret_cast<float*>(bases[B_GRADS] + reinterpret_cast<uintptr_t>(d.dst_b)) : nullptr;
            d.blk0 = blk; blk += d.nblk;
        }
        if (int r = launch_reduce(t, blk, st)) return r;
    }
    return 0;
}

int run_step(mi_unet& u, Step& s, char* const* bases, int training, uint64_t seed, hipStream_t st, unsigned* start_flag = nullptr, unsigned start_val = 0) {
    if (!bases_ok(s, bases)) { mi_set_error("step '" + s.layer + "' needs a buffer the caller did not pass"); return -1; }
    switch (s.kind) {
        case K_IGEMM:
        case K_KFINISH: {
            IgemmParams p = resolve(s, s.ig, bases);
            p.start_flag = start_flag; p.start_val = start_val;
            return s.kind == K_IGEMM ? launch_igemm(s.dtype, p, st) : launch_ksplit_finish(s.dtype, p, st);
        }
        case K_KFINISH_NORM: {
            NormParams p = resolve_with(s.rel_nm, s.nm, bases);
            p.seed = seed;
            norm_set_dropout(p, training ? u.cfg.dropout : 0.f);
            if (p.dy) { p.start_flag = start_flag; p.start_val = start_val; }
            return p.dy ? launch_finish_norm_bwd(s.dtype, resolve(s, s.ig, bases), p, st)
                        : launch_finish_norm(s.dtype, resolve(s, s.ig, bases), p, st);
        }
        case K_WGRAD: return launch_wgrad(s.dtype, resolve(s, s.wg, bases), st);
        case K_NORM_FWD:
        case K_NORM_BWD: {
            NormParams p = resolve(s, s.nm, bases);
            p.seed = seed;
            norm_set_dropout(p, training ? u.cfg.dropout : 0.f);
            if (s.kind == K_NORM_BWD) { p.start_flag = start_flag; p.start_val = start_val; }
            return s.kind == K_NORM_FWD ? launch_norm_fwd(s.dtype, p, st) : launch_norm_bwd(s.dtype, p, st);
        }
        case K_C1_FWD: return launch_c1_s2_fwd(s.dtype, resolve(s, s.c1, bases), st);
        case K_C1_T: return launch_c1_s2_T(s.dtype, resolve(s, s.c1, bases), st);
        case K_C1_WGRAD: return launch_c1_s2_wgrad(s.dtype, resolve(s, s.c1, bases), st);
        case K_STENCIL: {
            StencilParams p = resolve(s, s.stn, bases);
            if (p.nrm_stats) {
                NormParams d; memset(&d, 0, sizeof(d));
                norm_set_dropout(d, training ? u.cfg.dropout : 0.f);
                p.nrm_seed = seed; p.nrm_thresh = d.drop_thresh; p.nrm_scale = d.drop_scale;
            }
            return launch_stencil(p, st);
        }
        case K_STENCIL_WGRAD: return launch_stencil_wgrad(resolve(s, s.stn, bases), st);
        case K_ADD: {
            AddArgs a = resolve(s, s.add, bases);
            return launch_add(s.dtype, a.dst, a.dpitch, a.dcoff, a.src, a.spitch, a.scoff, a.c, a.npix, st);
        }
        case K_MEMSET: {
            MemsetArgs m = resolve(s, s.ms, bases);
            if (m.bytes) MI_HIP(hipMemsetAsync(m.ptr, 0, m.bytes, st));
            return 0;
        }
        case K_REDUCE: return run_reduce(u, s, bases, st);
        case K_PACK: return run_pack(u, s, bases, st);
    }
    mi_set_error("unknown step kind");
    return -1;
}

// Raises a start flag from a kernel of its own: used where the main stream has nothing left to enqueue behind the
// collected side-stream steps (end of the backward pass), so the kernel holds nobody back.
__global__ void raise_flag_kernel(unsigned* flag, unsigned val) {
    if (threadIdx.x == 0) __hip_atomic_store(flag, val, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
}

bool is_wgrad_kind(int kind) { return kind == K_WGRAD || kind == K_C1_WGRAD || kind == K_STENCIL_WGRAD; }

// The fork / join events order kernels of two streams of one device and are never inspected by the host.  Every
// kernel boundary already releases and acquires at device scope, so the events need no system-scope fence of
// their own; without it a record costs the main chain ~3 us instead of ~5 (30 records per training step).
// MI_EVENT_SYSTEM_FENCE=1 restores the default event behaviour.
unsigned fork_event_flags() {
    static const bool sys = getenv("MI_EVENT_SYSTEM_FENCE") && atoi(getenv("MI_EVENT_SYSTEM_FENCE")) != 0;
    return hipEventDisableTiming | (sys ? 0u : (unsigned)hipEventDisableSystemFence);
}

// Buffers of the later parts of a training step that must start at zero (loss sums, backward statistics): a forward
// pass that forks the side stream for the weight packing clears them there, off the main chain (each memset on the
// main stream is a ~5 us link of it); `done` tells the caller whether that happened.
struct PreClear { void* ptr[2]; size_t bytes[2]; bool done; };

// Optimizer step of one gradient bucket right behind that bucket's slab reduction (one-call training step on one
// device): a bucket's gradients are final there, and nothing later in the pass reads its parameters (the data-gradient
// kernels read the packed copies, or raw weights of the bucket's own layers, which precede its reduction), so only the
// last bucket's update is left for the end of the step.  `done`: bit b set once bucket b has been enqueued.
struct AdamHook { AdamParams p; unsigned done; };

int run_list(mi_unet& u, std::vector<Step>& list, int backward, int bucket, char* const* bases, int training, uint64_t seed,
             hipStream_t st, hipStream_t join_to = nullptr, PreClear* pre = nullptr, bool skip_memset = false,
             AdamHook* ah = nullptr) {
    // A second stream runs work that is off the critical chain: in the backward pass the weight gradients and
    // the slab reductions (joined at the end of the call), in the forward pass the weight packing (joined in
    // front of the first kernel that reads packed weights).  Timing runs stay on one stream.
    bool side = false;
    if (!u.prof) {
        if (u.use_side < 0) {
            u.use_side = getenv("MI_NO_SIDE_STREAM") ? 0 : 1;
            if (u.use_side) {
                if (hipStreamCreateWithFlags(&u.side, hipStreamNonBlocking) != hipSuccess) { (void)hipGetLastError(); u.use_side = 0; }
                else if (hipEventCreateWithFlags(&u.join_ev, fork_event_flags()) != hipSuccess) { (void)hipGetLastError(); u.use_side = 0; }
                else if (hipEventCreateWithFlags(&u.pack_ev, fork_event_flags()) != hipSuccess) { (void)hipGetLastError(); u.use_side = 0; }
            }
        }
        side = u.use_side == 1;
    }
    bool pending = false;
    bool pack_first_pending = false, pack_first_done = false;
    size_t nside = 0;
    // Weight gradients per fork of the side stream.  An event record plus a stream wait costs the host about as much
    // as three kernel launches; on a large batch every weight gradient forks on its own (it then starts beside the
    // kernels that suit it best: measured faster), on small workloads, where the host's enqueue time is what bounds
    // the step, several weight gradients share one fork (they are launched together behind the last one's producer:
    // nothing they read is ever overwritten, the workspace has no reuse).
    int fork_batch = 1;
    Step* deferred[16]; int ndeferred = 0; bool deferred_reduce = false;
    // Forks without a packet on the main stream.  Side-stream steps of the backward pass are collected; when the next
    // main-stream step is one whose first kernel can raise a start flag (the norm backward kernels), that step is
    // enqueued first, then the side stream gets a hipStreamWaitValue32 on the flag and the collected steps.  "That
    // kernel has started" implies everything enqueued before it has completed, which is what an event recorded in
    // front of it would have said, but a record holds the next kernel back by 2.5-5 us and releases the side stream
    // ~8 us after the producer ends instead of ~1 us (measured, scripts/micro/wait_value.hip).  The wait is always
    // enqueued AFTER the kernel that satisfies it, so two streams that share a hardware queue cannot deadlock.
    // A flag is raised to a value that changes with every call (compare: equal), so neither the previous pass's flags
    // nor the zeros the first step of every backward pass writes over the block (the flags live in the backward
    // statistics block) can release a wait early; a workspace this plan has not used before gets its flag words
    // cleared once, ordered in front of both streams.  In front of any other kind of step, and at the end of the
    // call, the collected steps are released by an event as before.  Bucket-by-bucket calls (data-parallel training)
    // use events only: two ranks time-sharing one GPU hung with the flag waits in place (cause not found; every rank of
    // a real job has a GPU of its own, but the event path is the proven one there).  MI_NO_FLAG_FORK=1: events only.
    unsigned* flag_base = nullptr;
    int nflag = 0;
    // Small workloads only (the criterion of the fork batching below): on a large batch the weight gradient that
    // starts ~8 us earlier shares more of the norm backward's HBM time than it gains (same box, flags / events:
    // headline 2.376 / 2.373 ms, 32-512 widths 5.39 / 5.31 ms; 8 x 256^2 0.80 / 0.87 ms, 96^3 patches 1.63 / 1.68 ms;
    // raised by the apply pass instead of the reduce pass: 2.44 ms).  MI_FLAG_FORK=1 / 0 forces the choice.
    static const int flag_env = getenv("MI_FLAG_FORK") ? atoi(getenv("MI_FLAG_FORK")) : -1;
    const bool flag_wanted = flag_env >= 0 ? flag_env != 0 : (long long)u.nb * u.rp[1] < (1ll << 20);
    if (side && backward && bucket < 0 && flag_wanted && bases[B_ARENA]) {
        static const bool off = getenv("MI_NO_FLAG_FORK") != nullptr;
        if (u.can_wait_value < 0) {
            int dev = 0, can = 0;
            if (hipGetDevice(&dev) != hipSuccess || hipDeviceGetAttribute(&can, hipDeviceAttributeCanUseStreamWaitValue, dev) != hipSuccess) { (void)hipGetLastError(); can = 0; }
            u.can_wait_value = can ? 1 : 0;
        }
        bool capturing = false;       // a stream under graph capture takes no value waits: events there
        {
            hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
            if (hipStreamIsCapturing(st, &cs) != hipSuccess) { (void)hipGetLastError(); capturing = true; }
            else capturing = cs != hipStreamCaptureStatusNone;
        }
        if (!off && !capturing && u.can_wait_value == 1) {
            flag_base = reinterpret_cast<unsigned*>(bases[B_ARENA] + u.off_flags);
            if (u.flag_seq == 0) u.flag_seq = 0x9e3779b9u * (unsigned)(reinterpret_cast<uintptr_t>(&u) >> 4) | 1u;
            u.flag_seq += 0x61c88647u;
            if (u.flag_seq == 0) u.flag_seq = 1;
            if (u.flag_arena != bases[B_ARENA]) {
                MI_HIP(hipMemsetAsync(flag_base, 0, (size_t)MI_NFLAG * 64, u.side));
                MI_HIP(hipEventRecord(u.join_ev, u.side));
                MI_HIP(hipStreamWaitEvent(st, u.join_ev, 0));
                u.flag_arena = bases[B_ARENA];
            }
        }
    }
    if (backward) {
        // event forks on small workloads: four weight gradients per fork; flag forks cost the chain nothing and
        // release every weight gradient on its own (measured: 8 x 256^2 0.798 / 0.803 / 0.805 ms at 1 / 2 / 4)
        static const int env = getenv("MI_FORK_BATCH") ? atoi(getenv("MI_FORK_BATCH")) : 0;
        fork_batch = env > 0 ? env : (!flag_base && (long long)u.nb * u.rp[1] < (1ll << 20) ? 4 : 1);
    }
    auto fork = [&]() -> int {      // the side stream waits for everything enqueued on the main stream so far
        if (nside >= u.side_ev.size()) {
            hipEvent_t e;
            MI_HIP(hipEventCreateWithFlags(&e, fork_event_flags()));
            u.side_ev.push_back(e);
        }
        hipEvent_t ready = u.side_ev[nside++];
        MI_HIP(hipEventRecord(ready, st));
        MI_HIP(hipStreamWaitEvent(u.side, ready, 0));
        return 0;
    };
    auto join = [&]() -> int {      // the main stream waits for everything enqueued on the side stream so far
        MI_HIP(hipEventRecord(u.join_ev, u.side));
        MI_HIP(hipStreamWaitEvent(st, u.join_ev, 0));
        pending = false;
        return 0;
    };
    auto join_other = [&](hipStream_t other) -> int {      // another stream of the caller waits instead; the main stream goes on
        MI_HIP(hipEventRecord(u.join_ev, u.side));
        MI_HIP(hipStreamWaitEvent(other, u.join_ev, 0));
        pending = false;
        return 0;
    };
    // Small workloads, whole pass in one call: the reduction of the second-to-last bucket leaves the side stream.  The
    // side stream ends the step with the weight gradients of the first level while the main stream has nothing left to
    // do, so that reduction runs there, behind the last main-stream step (an event recorded at its place in the side
    // stream's order says its weight gradients are complete).  Measured: 8 x 256^2 0.787 -> 0.779 ms per step, the
    // 96^3 patches and the headline batch unchanged; all reductions but the last moved: slower everywhere (0.784 ->
    // 0.807, 1.562 -> 1.577, 2.355 -> 2.375 ms).  Bucket-by-bucket calls keep every reduction where it is: the
    // caller's all-reduce of a bucket waits for it.
    static const bool late_off = getenv("MI_NO_LATE_REDUCE") != nullptr;
    const Step* last_reduce = nullptr; const Step* late_reduce = nullptr;
    if (backward && bucket < 0 && !late_off && (long long)u.nb * u.rp[1] < (1ll << 20))
        for (const Step& s : list) if (s.kind == K_REDUCE) { late_reduce = last_reduce; last_reduce = &s; }
    Step* late[8]; int nlate = 0;
    auto adam_bucket = [&](int b, hipStream_t s2) -> int {
        if (!ah || b < 0 || b >= MI_NUM_BUCKETS || ((ah->done >> b) & 1u)) return 0;
        ah->done |= 1u << b;
        if (u.bucket_cnt[b] <= 0) return 0;
        AdamParams q = ah->p;
        q.param += u.bucket_off[b]; q.grad += u.bucket_off[b]; q.m += u.bucket_off[b]; q.v += u.bucket_off[b];
        q.n = u.bucket_cnt[b];
        return launch_adamw(q, s2);
    };
    auto run_deferred = [&]() -> int {       // the collected steps, in order, on the side stream
        for (int k = 0; k < ndeferred; ++k) {
            Step* d = deferred[k];
            if (d == late_reduce && nlate < 8) {
                if ((size_t)nlate >= u.late_ev.size()) {
                    hipEvent_t ev;
                    MI_HIP(hipEventCreateWithFlags(&ev, fork_event_flags()));
                    u.late_ev.push_back(ev);
                }
                MI_HIP(hipEventRecord(u.late_ev[nlate], u.side));
                late[nlate++] = d;
                continue;
            }
            if (int r = run_step(u, *d, bases, training, seed, u.side)) return r;
            if (d->kind == K_REDUCE) { if (int r = adam_bucket(d->bucket, u.side)) return r; }
        }
        ndeferred = 0; deferred_reduce = false; pending = true;
        return 0;
    };
    auto run_late = [&]() -> int {           // on the main stream, behind its last step
        for (int k = 0; k < nlate; ++k) {
            MI_HIP(hipStreamWaitEvent(st, u.late_ev[k], 0));
            if (int r = run_step(u, *late[k], bases, training, seed, st)) return r;
            if (int r = adam_bucket(late[k]->bucket, st)) return r;
        }
        nlate = 0;
        return 0;
    };
    auto flush_by_event = [&]() -> int {     // one event fork for everything collected so far
        if (!ndeferred) return 0;
        if (int r = fork()) return r;
        return run_deferred();
    };
    auto body = [&]() -> int {
    for (size_t i = 0; i < list.size(); ++i) {
        Step& s = list[i];
        if (bucket >= 0 && s.bucket != bucket) continue;
        if (skip_memset && s.kind == K_MEMSET) continue;
        if (side && backward && (is_wgrad_kind(s.kind) || s.kind == K_REDUCE)) {
            if (ndeferred == 16) { if (int r = flush_by_event()) return r; }
            deferred[ndeferred++] = &s;
            if (s.kind == K_REDUCE) deferred_reduce = true;
            continue;
        }
        ProfEvent ev;
        if (u.prof) {
            MI_HIP(hipEventCreate(&ev.a)); MI_HIP(hipEventCreate(&ev.b));
            MI_HIP(hipEventRecord(ev.a, st));
        }
        if (side && !backward && s.kind == K_PACK) {
            if (!pack_first_pending) { if (int r = fork()) return r; }             // the second packing launch follows the first
            if (int r = run_step(u, s, bases, training, seed, u.side)) return r;
            if (!pack_first_pending && !pack_first_done) {
                MI_HIP(hipEventRecord(u.pack_ev, u.side));
                pack_first_pending = true;
            } else if (pre && !pre->done) {
                // behind the second packing launch: the first deep layer's join covers these too
                for (int k = 0; k < 2; ++k)
                    if (pre->ptr[k] && pre->bytes[k]) MI_HIP(hipMemsetAsync(pre->ptr[k], 0, pre->bytes[k], u.side));
                pre->done = true;
            }
            pending = true;
        } else {
            if (!backward && s.wait_pack && pack_first_pending) {
                MI_HIP(hipStreamWaitEvent(st, u.pack_ev, 0));
                pack_first_pending = false; pack_first_done = true;
            }
            if (!backward && s.join_side && pending) { if (int r = join()) return r; }
            // side-stream steps collected in front of this one are released now: several weight gradients share a
            // fork on small workloads (fork_batch), a gradient reduction never waits for more
            const bool release = ndeferred && (ndeferred >= fork_batch || deferred_reduce);
            if (release && flag_base && nflag < MI_NFLAG && (s.kind == K_NORM_BWD || s.kind == K_KFINISH_NORM || ((s.kind == K_IGEMM || s.kind == K_KFINISH) && !getenv("MI_NO_CONV_FLAG")))) {
                unsigned* sf = flag_base + (nflag++) * 16;
                if (int r = run_step(u, s, bases, training, seed, st, sf, u.flag_seq)) return r;
                MI_HIP(hipStreamWaitValue32(u.side, sf, u.flag_seq, hipStreamWaitValueEq, 0xffffffffu));
                if (int r = run_deferred()) return r;
            } else {
                if (release) { if (int r = flush_by_event()) return r; }
                if (int r = run_step(u, s, bases, training, seed, st)) return r;
                if (backward && s.kind == K_REDUCE) { if (int r = adam_bucket(s.bucket, st)) return r; }
            }
        }
        if (u.prof) {
            MI_HIP(hipEventRecord(ev.b, st));
            ev.backward = backward; ev.step = (int)i;
            u.events.push_back(ev);
        }
    }
    // Whole pass in one call, all but the smallest workloads: what is left behind the last main-stream step (the
    // weight gradient of the first layer and the last reduction) runs on the main stream, which has nothing else to
    // do: no fork in front of it and no join behind it (two event round trips of 9 us each at the end of the step).
    // The reduction reads slabs the side stream wrote, so the main stream waits for the side stream in front of it,
    // not in front of the weight gradient.  Measured: headline 2.357 -> 2.345 ms, 32-512 widths 5.339 -> 5.321 ms,
    // 96^3 patches 1.452 -> 1.445 ms; 8 x 256^2 0.758 -> 0.770 ms (there the late reduction runs beside an 11 us
    // weight gradient), hence the size criterion.  MI_NO_TAIL_ON_MAIN=1: the side stream as before.
    static const bool tail_off = getenv("MI_NO_TAIL_ON_MAIN") != nullptr;
    if (ndeferred && side && backward && bucket < 0 && !tail_off && (long long)u.nb * u.rp[1] >= (1ll << 18)) {
        bool joined = false;
        for (int k = 0; k < ndeferred; ++k) {
            Step* d = deferred[k];
            if (d->kind == K_REDUCE && !joined) { if (pending) { if (int r = join()) return r; } joined = true; }
            if (int r = run_step(u, *d, bases, training, seed, st)) return r;
            if (d->kind == K_REDUCE) { if (int r = adam_bucket(d->bucket, st)) return r; }
        }
        ndeferred = 0; deferred_reduce = false;
        return 0;
    }
    if (ndeferred && flag_base && nflag < MI_NFLAG) {
        unsigned* sf = flag_base + (nflag++) * 16;
        hipLaunchKernelGGL(raise_flag_kernel, dim3(1), dim3(64), 0, st, sf, u.flag_seq);
        MI_HIP(hipGetLastError());
        MI_HIP(hipStreamWaitValue32(u.side, sf, u.flag_seq, hipStreamWaitValueEq, 0xffffffffu));
        return run_deferred();
    }
    return flush_by_event();
    };
    if (int r = body()) return r;
    if (int r = run_late()) return r;
    if (join_to && side) return join_other(join_to);       // also when nothing is pending: earlier calls' work may still run
    if (pending) { if (int r = join()) return r; }
    return 0;
}

}  // namespace

extern "C" {

int mi_unet_create(const mi_unet_cfg* cfg, mi_unet** out) {
    MI_CHECK(cfg && out, "null argument");
    MI_CHECK(cfg->spatial_dims == 2 || cfg->spatial_dims == 3, "spatial_dims must be 2 or 3");
    MI_CHECK(cfg->in_channels == 1 && cfg->out_channels == 1, "the HIP path implements in_channels = out_channels = 1");
    MI_CHECK(cfg->n_levels >= 2 && cfg->n_levels <= 8, "2 <= len(channels) <= 8");
    MI_CHECK(cfg->num_res_units >= 1 && cfg->num_res_units <= 4, "1 <= num_res_units <= 4");
    MI_CHECK(cfg->dropout >= 0.f && cfg->dropout < 1.f, "0 <= dropout < 1");
    MI_CHECK(cfg->dtype == 0 || cfg->dtype == 1, "dtype must be 0 (fp32) or 1 (bf16)");
    MI_CHECK(cfg->batch >= 1, "batch must be positive");
    for (int i = 0; i < cfg->n_levels; ++i) MI_CHECK(cfg->channels[i] % 16 == 0 && cfg->channels[i] >= 16 && cfg->channels[i] <= 1024, "channels must be multiples of 16 in [16, 1024]");
    for (int i = 0; i < cfg->n_levels - 1; ++i) MI_CHECK(cfg->strides[i] == 2, "strides must all be 2");
    MI_CHECK(cfg->channels[0] <= 128, "channels[0] must be at most 128");
    const int div = 1 << (cfg->n_levels - 1);
    for (int d = 0; d < 3; ++d) {
        if (cfg->spatial_dims == 2 && d == 0) { MI_CHECK(cfg->dims[0] == 1, "dims[0] must be 1 in 2-D"); continue; }
        MI_CHECK(cfg->dims[d] >= div && cfg->dims[d] % div == 0, "every spatial size must be divisible by 2^(len(channels)-1)");
    }
    mi_unet* u = new mi_unet();
    u->cfg = *cfg;
    u->L = cfg->n_levels; u->R = cfg->num_res_units; u->nd = cfg->spatial_dims; u->dt = cfg->dtype;
    u->esz = cfg->dtype ? 2 : 4; u->nb = cfg->batch;
    u->KK = cfg->spatial_dims == 3 ? 27 : 9;
    for (int r = 0; r < u->L; ++r) {
        u->rp[r] = 1;
        for (int d = 0; d < 3; ++d) {
            u->rd[r][d] = (cfg->spatial_dims == 2 && d == 0) ? 1 : cfg->dims[d] >> r;
            u->rp[r] *= u->rd[r][d];
        }
    }
    {
        const long long big = (long long)u->nb * u->rp[0] * 2 * cfg->channels[0];
        if (big > 2000000000LL) { delete u; mi_set_error("mi_unet_create: batch * pixels too large for 32-bit pixel indices"); return -1; }
    }
    enum_params(*u);
    if (!build(*u)) { delete u; return -1; }
    *out = u;
    return 0;
}

void mi_unet_destroy(mi_unet* u) {
    if (!u) return;
    if (u->host_calls) fprintf(stderr, "[libunet_mi355] host time inside mi_unet_train_step: %.1f us per call over %lld calls\n", u->host_ns / 1e3 / u->host_calls, u->host_calls);
    for (ProfEvent& e : u->events) { (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b); }
    for (hipEvent_t e : u->side_ev) (void)hipEventDestroy(e);
    if (u->join_ev) (void)hipEventDestroy(u->join_ev);
    for (hipEvent_t ev : u->late_ev) (void)hipEventDestroy(ev);
    if (u->pack_ev) (void)hipEventDestroy(u->pack_ev);
    if (u->side) (void)hipStreamDestroy(u->side);
    delete u;
}

int mi_unet_num_params(const mi_unet* u) { return (int)u->params.size(); }
int mi_unet_num_buckets(const mi_unet*) { return MI_NUM_BUCKETS; }

int mi_unet_param_info(const mi_unet* u, int i, mi_param_info* info) {
    MI_CHECK(u && info && i >= 0 && i < (int)u->params.size(), "bad index");
    const ParamRec& r = u->params[i];
    memset(info, 0, sizeof(*info));
    snprintf(info->name, sizeof(info->name), "%s", r.name.c_str());
    info->ndim = r.ndim;
    for (int k = 0; k < 5; ++k) info->shape[k] = r.shape[k];
    info->numel = r.numel; info->offset = r.offset; info->bucket = r.bucket;
    return 0;
}

int64_t mi_unet_param_floats(const mi_unet* u) { return u->nfloats; }
size_t mi_unet_workspace_bytes(const mi_unet* u) { return u->arena; }

int mi_unet_bucket_range(const mi_unet* u, int bucket, int64_t* offset, int64_t* count) {
    MI_CHECK(u && offset && count && bucket >= 0 && bucket < MI_NUM_BUCKETS, "bad bucket");
    *offset = u->bucket_off[bucket]; *count = u->bucket_cnt[bucket];
    return 0;
}

double mi_unet_flops(const mi_unet* u, int backward) { return backward ? u->flops_bwd : u->flops_fwd; }
int mi_unet_num_steps(const mi_unet* u, int backward) { return (int)(backward ? u->bwd.size() : u->fwd.size()); }

int mi_unet_forward(mi_unet* u, const float* params, const float* x, float* logits, void* workspace, int training,
                    uint64_t seed, void* stream) {
    MI_CHECK(u && params && x && logits && workspace, "null argument");
    char* bases[B_NBASE] = {nullptr};
    bases[B_ARENA] = (char*)workspace; bases[B_PARAMS] = (char*)params; bases[B_XIN] = (char*)x; bases[B_LOGITS] = (char*)logits;
    hipStream_t st = (hipStream_t)stream;
    return run_list(*u, u->fwd, 0, -1, bases, training, seed, st);
}

int mi_unet_backward(mi_unet* u, const float* params, const float* x, const float* dlogits, float* grads, void* workspace,
                     int training, uint64_t seed, int bucket, void* stream) {
    MI_CHECK(u && params && x && dlogits && grads && workspace, "null argument");
    MI_CHECK(u->cfg.with_backward, "plan was created without a backward pass");
    MI_CHECK(bucket < MI_NUM_BUCKETS, "bucket must be < MI_NUM_BUCKETS");
    char* bases[B_NBASE] = {nullptr};
    bases[B_ARENA] = (char*)workspace; bases[B_PARAMS] = (char*)params; bases[B_GRADS] = (char*)grads;
    bases[B_DLOGITS] = (char*)dlogits; bases[B_XIN] = (char*)x;
    hipStream_t st = (hipStream_t)stream;
    return run_list(*u, u->bwd, 1, bucket, bases, training, seed, st);
}

int mi_unet_backward_bucket(mi_unet* u, const float* params, const float* x, const float* dlogits, float* grads, void* workspace,
                            int training, uint64_t seed, int bucket, void* stream, void* ready_stream) {
    MI_CHECK(u && params && x && dlogits && grads && workspace, "null argument");
    MI_CHECK(u->cfg.with_backward, "plan was created without a backward pass");
    MI_CHECK(bucket >= 0 && bucket < MI_NUM_BUCKETS, "bucket must be in [0, MI_NUM_BUCKETS)");
    char* bases[B_NBASE] = {nullptr};
    bases[B_ARENA] = (char*)workspace; bases[B_PARAMS] = (char*)params; bases[B_GRADS] = (char*)grads;
    bases[B_DLOGITS] = (char*)dlogits; bases[B_XIN] = (char*)x;
    return run_list(*u, u->bwd, 1, bucket, bases, training, seed, (hipStream_t)stream, (hipStream_t)ready_stream);
}

static int train_step_body(mi_unet* u, const mi_train_step_args* a, void* stream);
int mi_unet_train_step(mi_unet* u, const mi_train_step_args* a, void* stream) {
    static const bool timing = getenv("MI_HOST_TIMING") != nullptr;
    if (!timing) return train_step_body(u, a, stream);
    const auto t0 = std::chrono::steady_clock::now();
    const int r = train_step_body(u, a, stream);
    if (u) { u->host_ns += std::chrono::duration<double, std::nano>(std::chrono::steady_clock::now() - t0).count(); u->host_calls += 1; }
    return r;
}
static int train_step_body(mi_unet* u, const mi_train_step_args* a, void* stream) {
    MI_CHECK(u && a, "null argument");
    MI_CHECK(a->params && a->x && a->target && a->logits && a->dlogits && a->loss_sums && a->loss && a->grads && a->workspace,
             "null buffer");
    MI_CHECK(u->cfg.with_backward, "plan was created without a backward pass");
    MI_CHECK(!a->do_adamw || (a->exp_avg && a->exp_avg_sq && a->step >= 1), "AdamW needs its moment buffers and a step count >= 1");
    const long long npix = u->rp[0];
    MI_CHECK(npix < 2147483647LL, "bad sizes");
    hipStream_t st = (hipStream_t)stream;
    char* bases[B_NBASE] = {nullptr};
    bases[B_ARENA] = (char*)a->workspace; bases[B_PARAMS] = (char*)a->params; bases[B_XIN] = (char*)a->x;
    bases[B_LOGITS] = (char*)a->logits;
    // loss sums and backward statistics are zeroed on the side stream during the forward pass when it has one
    PreClear pre; memset(&pre, 0, sizeof(pre));
    static const bool no_pre = getenv("MI_NO_PRECLEAR") != nullptr;
    if (!no_pre) {
        pre.ptr[0] = a->loss_sums; pre.bytes[0] = (size_t)u->nb * 4 * sizeof(double);
        int nms = 0;
        for (Step& s : u->bwd)
            if (s.kind == K_MEMSET) {
                MemsetArgs m = resolve(s, s.ms, bases);
                pre.ptr[1] = m.ptr; pre.bytes[1] = m.bytes; ++nms;
            }
        if (nms != 1) { pre.ptr[0] = pre.ptr[1] = nullptr; }       // unexpected plan shape: the ordinary memsets stay
    }
    if (int r = run_list(*u, u->fwd, 0, -1, bases, 1, a->seed, st, nullptr, pre.ptr[0] ? &pre : nullptr)) return r;
    LossParams p; memset(&p, 0, sizeof(p));
    p.sums_cleared = pre.done ? 1 : 0;
    static const bool no_fuse = getenv("MI_NO_LOSS_FUSE") != nullptr;
    p.fuse_finalize = no_fuse ? 0 : 1;
    p.logits = a->logits; p.target = a->target; p.tdtype = a->target_dtype; p.fold2 = a->fold2; p.sums = a->loss_sums;
    p.loss = a->loss; p.loss_accum = a->loss_accum; p.nb = u->nb; p.npix = (int)npix;
    p.smooth_nr = a->smooth_nr; p.smooth_dr = a->smooth_dr; p.lambda_dice = a->lambda_dice; p.lambda_ce = a->lambda_ce;
    if (int r = launch_loss_fwd(p, st)) return r;
    p.dlogits = a->dlogits; p.gout = nullptr; p.gscale = 1.0f;
    if (int r = launch_loss_bwd(p, st)) return r;
    bases[B_LOGITS] = nullptr; bases[B_GRADS] = (char*)a->grads; bases[B_DLOGITS] = (char*)a->dlogits;
    // large workloads only: three more launches do not pay where the step is bound by launches (measured: headline
    // 2.357 -> 2.344 ms, 96^3 patches unchanged, 8 x 256^2 0.777 -> 0.779 ms)
    static const bool no_hook = getenv("MI_NO_BUCKET_ADAMW") != nullptr;
    if (a->do_adamw && !no_hook && (long long)u->nb * u->rp[1] >= (1ll << 20)) {
        MI_CHECK((((uintptr_t)a->params | (uintptr_t)a->grads | (uintptr_t)a->exp_avg | (uintptr_t)a->exp_avg_sq) & 15) == 0, "buffers must be 16-byte aligned");
        AdamHook ah; memset(&ah, 0, sizeof(ah));
        AdamParams& q = ah.p;
        q.param = a->params; q.grad = a->grads; q.m = a->exp_avg; q.v = a->exp_avg_sq; q.n = 0;
        q.decay = 1.0f - a->lr * a->weight_decay; q.beta1 = a->beta1; q.beta2 = a->beta2; q.eps = a->eps;
        const double bc1 = 1.0 - pow((double)a->beta1, (double)a->step), bc2 = 1.0 - pow((double)a->beta2, (double)a->step);
        q.step_size = (float)((double)a->lr / bc1); q.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2)); q.gscale = a->grad_scale;
        if (int r = run_list(*u, u->bwd, 1, -1, bases, 1, a->seed, st, nullptr, nullptr, pre.done, &ah)) return r;
        for (int b = 0; b < MI_NUM_BUCKETS; ++b)       // buckets whose reduction the pass did not see (none in a regular plan)
            if (!((ah.done >> b) & 1u) && u->bucket_cnt[b] > 0) {
                AdamParams t = q;
                t.param += u->bucket_off[b]; t.grad += u->bucket_off[b]; t.m += u->bucket_off[b]; t.v += u->bucket_off[b];
                t.n = u->bucket_cnt[b];
                if (int r = launch_adamw(t, st)) return r;
            }
        return 0;
    }
    if (int r = run_list(*u, u->bwd, 1, -1, bases, 1, a->seed, st, nullptr, nullptr, pre.done)) return r;
    if (a->do_adamw)
        return mi_adamw_step(a->params, a->grads, a->exp_avg, a->exp_avg_sq, u->nfloats, a->lr, a->beta1, a->beta2, a->eps,
                             a->weight_decay, a->step, a->grad_scale, stream);
    return 0;
}

// debug aid: arena offset of the tensor each step writes (-1 when it writes outside the arena)
int64_t mi_unet_step_output_offset(const mi_unet* u, int backward, int i) {
    const std::vector<Step>& l = backward ? u->bwd : u->fwd;
    if (i < 0 || i >= (int)l.size()) return -1;
    const Step& s = l[i];
    auto arena_field = [&](size_t off) {
        for (const Reloc& r : s.rel) if (r.field_off == off && r.base == B_ARENA) return true;
        return false;
    };
    switch (s.kind) {
        case K_IGEMM: return arena_field(offsetof(IgemmParams, out)) ? (int64_t)(uintptr_t)s.ig.out : -1;
        case K_NORM_FWD: case K_NORM_BWD: return arena_field(offsetof(NormParams, out)) ? (int64_t)(uintptr_t)s.nm.out : -1;
        case K_KFINISH_NORM: return (int64_t)(uintptr_t)s.nm.out;
        case K_C1_FWD: return arena_field(offsetof(C1Params, y_out)) ? (int64_t)(uintptr_t)s.c1.y_out : -1;
        case K_C1_T: return arena_field(offsetof(C1Params, x1_out)) ? (int64_t)(uintptr_t)s.c1.x1_out : -1;
        case K_STENCIL: return arena_field(offsetof(StencilParams, out)) ? (int64_t)(uintptr_t)s.stn.out : -1;
        case K_WGRAD: return (int64_t)(uintptr_t)s.wg.partial;
        case K_C1_WGRAD: return (int64_t)(uintptr_t)s.c1.partial;
        case K_STENCIL_WGRAD: return (int64_t)(uintptr_t)s.stn.partial;
        default: return -1;
    }
}

// debug aid: arena offset of the backward-statistics block of a norm backward step (-1 otherwise)
int64_t mi_unet_step_aux_offset(const mi_unet* u, int backward, int i) {
    const std::vector<Step>& l = backward ? u->bwd : u->fwd;
    if (i < 0 || i >= (int)l.size()) return -1;
    const Step& s = l[i];
    if (s.kind == K_NORM_BWD) return (int64_t)(uintptr_t)s.nm.bstats;
    if (s.kind == K_KFINISH_NORM) return (int64_t)(uintptr_t)(backward ? (const void*)s.nm.bstats : (const void*)s.nm.stats);
    if (s.kind == K_NORM_FWD) return (int64_t)(uintptr_t)s.nm.stats;
    return -1;
}

int mi_unet_num_norms(const mi_unet* u) { return (int)u->norms.size(); }
int mi_unet_norm_info(const mi_unet* u, int k, mi_norm_info* info) {
    MI_CHECK(u && info && k >= 0 && k < (int)u->norms.size(), "bad index");
    *info = u->norms[k];
    return 0;
}

int mi_unet_prof_enable(mi_unet* u, int on) { u->prof = on != 0; return 0; }

int mi_unet_prof_collect(mi_unet* u) {
    for (ProfEvent& e : u->events) {
        MI_HIP(hipEventSynchronize(e.b));
        float ms = 0.f;
        MI_HIP(hipEventElapsedTime(&ms, e.a, e.b));
        Step& s = e.backward ? u->bwd[e.step] : u->fwd[e.step];
        s.total_ms += ms; s.launches += 1;
        (void)hipEventDestroy(e.a); (void)hipEventDestroy(e.b);
    }
    u->events.clear();
    return 0;
}

int mi_unet_prof_reset(mi_unet* u) {
    for (Step& s : u->fwd) { s.total_ms = 0; s.launches = 0; }
    for (Step& s : u->bwd) { s.total_ms = 0; s.launches = 0; }
    return 0;
}

int mi_unet_step_info(const mi_unet* u, int backward, int i, mi_step_info* info) {
    const std::vector<Step>& l = backward ? u->bwd : u->fwd;
    MI_CHECK(i >= 0 && i < (int)l.size() && info, "bad index");
    const Step& s = l[i];
    memset(info, 0, sizeof(*info));
    snprintf(info->layer, sizeof(info->layer), "%s", s.layer.c_str());
    snprintf(info->kernel, sizeof(info->kernel), "%s", s.kernel.c_str());
    info->work = s.work; info->work_is_flops = s.work_is_flops; info->total_ms = s.total_ms; info->launches = s.launches;
    info->bytes = s.work_is_flops ? s.bytes : s.work;
    return 0;
}

// ---------------- loss / metric / optimizer / sliding window ----------------
int mi_dice_ce_loss_fwd(const float* logits, const void* target, int target_dtype, int fold2, int nb, int64_t npix,
                        float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce, double* sums, float* loss,
                        float* loss_accum, void* stream) {
    MI_CHECK(logits && target && sums && loss, "null argument");
    MI_CHECK(npix > 0 && npix < 2147483647LL && nb > 0, "bad sizes");
    LossParams p; memset(&p, 0, sizeof(p));
    p.logits = logits; p.target = target; p.tdtype = target_dtype; p.fold2 = fold2; p.sums = sums; p.loss = loss;
    p.loss_accum = loss_accum; p.nb = nb; p.npix = (int)npix; p.smooth_nr = smooth_nr; p.smooth_dr = smooth_dr;
    p.lambda_dice = lambda_dice; p.lambda_ce = lambda_ce;
    return launch_loss_fwd(p, (hipStream_t)stream);
}

int mi_dice_ce_loss_bwd(const float* logits, const void* target, int target_dtype, int fold2, int nb, int64_t npix,
                        float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce, const double* sums,
                        const float* grad_out, float grad_scale, float* dlogits, void* stream) {
    MI_CHECK(logits && target && sums && dlogits, "null argument");
    MI_CHECK(npix > 0 && npix < 2147483647LL && nb > 0, "bad sizes");
    LossParams p; memset(&p, 0, sizeof(p));
    p.logits = logits; p.target = target; p.tdtype = target_dtype; p.fold2 = fold2; p.sums = const_cast<double*>(sums);
    p.dlogits = dlogits; p.gout = grad_out; p.gscale = grad_scale; p.nb = nb; p.npix = (int)npix;
    p.smooth_nr = smooth_nr; p.smooth_dr = smooth_dr; p.lambda_dice = lambda_dice; p.lambda_ce = lambda_ce;
    return launch_loss_bwd(p, (hipStream_t)stream);
}

int mi_dice_counts(const float* pred, int pred_is_binary, float thresh, const void* target, int target_dtype, int fold2,
                   int nb, int64_t npix, uint64_t* counts, void* stream) {
    MI_CHECK(pred && target && counts, "null argument");
    MI_CHECK(npix > 0 && npix < 2147483647LL && nb > 0, "bad sizes");
    CountParams p; memset(&p, 0, sizeof(p));
    p.pred = pred; p.pred_is_binary = pred_is_binary; p.thresh = thresh; p.target = target; p.tdtype = target_dtype;
    p.fold2 = fold2; p.counts = (unsigned long long*)counts; p.nb = nb; p.npix = (int)npix;
    return launch_dice_counts(p, (hipStream_t)stream);
}

int mi_fold_labels(void* target, int target_dtype, int64_t n, void* stream) {
    MI_CHECK(target && n >= 0, "bad argument");
    return launch_fold_labels(target, target_dtype, n, (hipStream_t)stream);
}

int mi_adamw_step(float* params, const float* grads, float* exp_avg, float* exp_avg_sq, int64_t n, float lr, float beta1,
                  float beta2, float eps, float weight_decay, int64_t step, float grad_scale, void* stream) {
    MI_CHECK(params && grads && exp_avg && exp_avg_sq && n > 0 && step >= 1, "bad argument");
    MI_CHECK((((uintptr_t)params | (uintptr_t)grads | (uintptr_t)exp_avg | (uintptr_t)exp_avg_sq) & 15) == 0, "buffers must be 16-byte aligned");
    AdamParams p;
    p.param = params; p.grad = grads; p.m = exp_avg; p.v = exp_avg_sq; p.n = n;
    p.decay = 1.0f - lr * weight_decay; p.beta1 = beta1; p.beta2 = beta2; p.eps = eps;
    const double bc1 = 1.0 - pow((double)beta1, (double)step), bc2 = 1.0 - pow((double)beta2, (double)step);
    p.step_size = (float)((double)lr / bc1); p.inv_bc2_sqrt = (float)(1.0 / sqrt(bc2)); p.gscale = grad_scale;
    return launch_adamw(p, (hipStream_t)stream);
}

int mi_sw_gather(const float* image, float* windows, const int32_t* starts, int nwin, int win0, int nwin_launch, int nb,
                 const int32_t img[3], const int32_t roi[3], const int32_t pad_lo[3], float cval, void* stream) {
    MI_CHECK(image && windows && starts && nwin > 0 && nwin_launch > 0, "bad argument");
    SwParams p; memset(&p, 0, sizeof(p));
    p.image = image; p.windows = windows; p.starts = starts; p.nwin = nwin; p.win0 = win0; p.nb = nb; p.cval = cval;
    for (int d = 0; d < 3; ++d) { p.img[d] = img[d]; p.roi[d] = roi[d]; p.pad_lo[d] = pad_lo[d]; }
    return launch_sw_gather(p, nwin_launch, (hipStream_t)stream);
}

int mi_sw_blend(const float* win_out, float* out, const int32_t* starts, int nwin, int nb, const int32_t img[3],
                const int32_t roi[3], const int32_t pad_lo[3], void* stream) {
    MI_CHECK(win_out && out && starts && nwin > 0, "bad argument");
    SwParams p; memset(&p, 0, sizeof(p));
    p.win_out = win_out; p.out = out; p.starts = starts; p.nwin = nwin; p.nb = nb;
    for (int d = 0; d < 3; ++d) { p.img[d] = img[d]; p.roi[d] = roi[d]; p.pad_lo[d] = pad_lo[d]; }
    return launch_sw_blend(p, (hipStream_t)stream);
}

}  // extern "C"
