// Instance-norm (no affine, biased variance) + dropout + PReLU, forward and backward, on
// channels-last tensors.  Bandwidth-bound: one 16-byte vector per lane per access, the
// per-(image, channel) statistics come from the producing convolution's epilogue
// (forward) or from a reduction pass (backward).
//
// forward :  xh = (x - mean) * rstd ; d = keep ? xh * scale : 0 ; y = d > 0 ? d : alpha*d ; out = y (+ res)
// backward:  gd = dy * (d > 0 ? 1 : alpha) ; gh = keep ? gd * scale : 0
//            dx = rstd * (gh - mean_p(gh) - xh * mean_p(gh * xh)) ;  dalpha = sum dy * min(d, 0)
#include "common.h"
#include "kparams.h"

namespace {

// mean / rstd of every channel of image `img` into LDS (2 floats per channel); the sums are fp64
__device__ __forceinline__ void load_moments(const double* stats, int img, int c, int npix, float eps, float* sm) {
    for (int ch = threadIdx.x; ch < c; ch += blockDim.x) {
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int sl = 0; sl < MI_NSLOT; ++sl) {
            const double* q = stats + (((size_t)img * MI_NSLOT + sl) * c + ch) * 2;
            s1 += q[0]; s2 += q[1];
        }
        const double mean = s1 / (double)npix;
        double var = s2 / (double)npix - mean * mean;
        var = var > 0.0 ? var : 0.0;
        sm[ch * 2] = (float)mean;
        sm[ch * 2 + 1] = (float)(1.0 / sqrt(var + (double)eps));
    }
}

#ifndef NORM_U_N
#define NORM_U_N 4
#endif
constexpr int NORM_U = NORM_U_N;      // independent 16-byte loads in flight per thread and tensor (forward)
#ifndef NORM_UB_N
#define NORM_UB_N 2
#endif
constexpr int NORM_UB = NORM_UB_N;   // the same for the backward kernels (two input tensors)

// Thread layout shared by the three kernels: cv_n = 16-byte vectors per pixel, a sweep of the workgroup
// covers rows = 256 / cv_n pixels, thread (prow, cv) keeps the same channel group for all its pixels, so
// the moments of its channels live in registers.  Workgroup b handles pixels [b*ppb, (b+1)*ppb).
struct NormLane {
    int cv, prow, rows, ch0, pbeg, pend;
    bool active;
};
template <int VEC>
__device__ __forceinline__ NormLane norm_lane(const NormParams& p) {
    NormLane l;
    const int cv_n = p.c / VEC;
    l.rows = 256 / cv_n;
    l.cv = threadIdx.x % cv_n; l.prow = threadIdx.x / cv_n;
    l.ch0 = l.cv * VEC;
    const int ppb = (p.npix + gridDim.x - 1) / gridDim.x;
    l.pbeg = blockIdx.x * ppb;
    l.pend = min(p.npix, l.pbeg + ppb);
    l.active = l.prow < l.rows;
    return l;
}

// Sums the per-thread partials s[2*VEC] (sums of gh, then of gh*xh, for the thread's VEC channels) over all pixel
// rows of the workgroup; on return part[ccv * 2*VEC + e] holds the totals of channel vector ccv (after the final
// barrier).  Threads of one channel vector sit cv_n lanes apart: when cv_n is a power of two that fits a wave, the
// rows inside a wave are folded with lane exchanges and only one value set per wave goes through LDS (one
// barrier instead of one per tree level); other channel counts use a fixed-shape LDS tree.  All orders are fixed.
template <int O>
__device__ __forceinline__ float row_shr_add(float v) {      // v + (value of the lane O below in the 16-lane DPP row, 0 past its start)
    return v + __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x110 + O, 0xf, 0xf, true));
}
template <int VEC>
__device__ __forceinline__ void sum_rows(float* s, float* part, const NormLane& l, int cv_n) {
    constexpr int W = 2 * VEC;
    if (cv_n == 1 || cv_n == 2 || cv_n == 4) {
        // few channel vectors: the threads of one vector sit 1, 2 or 4 lanes apart inside a 16-lane DPP row; row
        // shifts (register-to-register) fold the row into its last cv_n lanes, the 16 rows of the workgroup
        // then meet in LDS
#pragma unroll
        for (int e = 0; e < W; ++e) {
            float v = s[e];
            if (cv_n == 1) v = row_shr_add<1>(v);
            if (cv_n <= 2) v = row_shr_add<2>(v);
            v = row_shr_add<4>(v);
            v = row_shr_add<8>(v);
            s[e] = v;
        }
        const int lr16 = threadIdx.x & 15, row = threadIdx.x >> 4;
        if (lr16 >= 16 - cv_n) {
#pragma unroll
            for (int e = 0; e < W; ++e) part[(row * cv_n + (lr16 & (cv_n - 1))) * W + e] = s[e];
        }
        __syncthreads();
        const int n = cv_n * W;
        for (int i = threadIdx.x; i < n; i += 256) {
            float tot = part[i];
#pragma unroll
            for (int r = 1; r < 16; ++r) tot += part[r * n + i];
            part[i] = tot;                                   // column i is touched by this thread only
        }
        __syncthreads();
    } else if ((cv_n & (cv_n - 1)) == 0 && cv_n <= 64 && cv_n >= 8) {     // at most three exchange steps; below that the tree is cheaper
        for (int o = 32; o >= cv_n; o >>= 1) {
#pragma unroll
            for (int e = 0; e < W; ++e) s[e] += __shfl_xor(s[e], o, 64);
        }
        const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
        if (lane < cv_n) {
#pragma unroll
            for (int e = 0; e < W; ++e) part[(w * cv_n + lane) * W + e] = s[e];
        }
        __syncthreads();
        const int n = cv_n * W;
        for (int i = threadIdx.x; i < n; i += 256) {
            const float tot = ((part[i] + part[i + n]) + part[i + 2 * n]) + part[i + 3 * n];
            part[i] = tot;                                   // column i is touched by this thread only
        }
        __syncthreads();
    } else {
#pragma unroll
        for (int e = 0; e < W; ++e) part[threadIdx.x * W + e] = s[e];
        __syncthreads();
        for (int n = l.rows; n > 1;) {
            const int half = (n + 1) >> 1;
            if (l.prow + half < n) {
#pragma unroll
                for (int e = 0; e < W; ++e) part[threadIdx.x * W + e] += part[(threadIdx.x + half * cv_n) * W + e];
            }
            __syncthreads();
            n = half;
        }
    }
}

// The three streaming kernels below share one shape: a thread walks its pixels NORM_U at a time, all loads of a
// step are issued before the first use, full steps carry no bounds checks (a short tail loop takes the rest),
// and dropout / residual are compile-time switches, so the inner loops are straight-line code.
template <class TR, bool DROP, bool RES>
__global__ __launch_bounds__(256) void norm_fwd_kernel(const NormParams p) {
    constexpr int VEC = TR::VEC, ESZ = TR::ESZ;
    extern __shared__ __attribute__((aligned(16))) float sm[];
    const int img = blockIdx.y;
    load_moments(p.stats, img, p.c, p.npix, p.eps, sm);
    __syncthreads();
    const NormLane l = norm_lane<VEC>(p);
    if (!l.active) return;
    const float alpha = *p.alpha;
    const DropKeys dk = dropout_keys(p.seed, p.layer, img, p.drop_thresh);
    const unsigned cq = p.c / 4, gq = l.ch0 / 4;    // dropout groups per pixel, first group of this lane
    const float scale = p.drop_scale;
    float mean[VEC], rstd[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mean[e] = sm[(l.ch0 + e) * 2]; rstd[e] = sm[(l.ch0 + e) * 2 + 1]; }
    const size_t img_pix = (size_t)img * p.npix;
    const char* xb = p.x + (img_pix * p.x_pitch + p.x_coff) * ESZ;
    const char* rb = RES ? p.res + (img_pix * p.res_pitch + p.res_coff) * ESZ : nullptr;
    char* ob = p.out + (img_pix * p.out_pitch + p.out_coff) * ESZ;
    const unsigned xs = p.x_pitch * ESZ, rs = p.res_pitch * ESZ, os = p.out_pitch * ESZ, cb = l.ch0 * ESZ;   // 32-bit offsets inside one image
    auto unit = [&](const uint4& xr, const uint4& rr, int pix) {
        float x[VEC], r[VEC], o[VEC];
        bool keep[VEC];
        unpack_vec<TR>(xr, x);
        if (RES) unpack_vec<TR>(rr, r);
        if (DROP) {
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) dropout_keep4(dk, (unsigned)pix * cq + gq + q, keep + 4 * q);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            float d = (x[e] - mean[e]) * rstd[e];
            if (DROP) d = keep[e] ? d * scale : 0.f;
            float y = d > 0.f ? d : alpha * d;
            if (RES) y += r[e];
            o[e] = y;
        }
        store_vec<TR>(ob + ((unsigned)pix * os + cb), o);
    };
    int pix0 = l.pbeg + l.prow;
    for (; pix0 + (NORM_U - 1) * l.rows < l.pend; pix0 += l.rows * NORM_U) {
        uint4 xr[NORM_U], rr[NORM_U];
#pragma unroll
        for (int u = 0; u < NORM_U; ++u) {
            const int pix = pix0 + u * l.rows;
            xr[u] = *reinterpret_cast<const uint4*>(xb + ((unsigned)pix * xs + cb));
            if (RES) rr[u] = *reinterpret_cast<const uint4*>(rb + ((unsigned)pix * rs + cb));
        }
#pragma unroll
        for (int u = 0; u < NORM_U; ++u) { unit(xr[u], rr[u], pix0 + u * l.rows); __builtin_amdgcn_sched_barrier(0); }
    }
    for (; pix0 < l.pend; pix0 += l.rows) {
        uint4 xr = *reinterpret_cast<const uint4*>(xb + ((unsigned)pix0 * xs + cb)), rr = xr;
        if (RES) rr = *reinterpret_cast<const uint4*>(rb + ((unsigned)pix0 * rs + cb));
        unit(xr, rr, pix0);
    }
}

// backward pass 1: per-(image, channel) sums of gh and gh*xh, and the slope gradient.
// Reduction order is fixed inside a workgroup (per-thread partials, then sum_rows);
// workgroup sums go to fp64 accumulators.
template <class TR, bool DROP>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(4))) void norm_bwd_reduce_kernel(const NormParams p) {
    constexpr int VEC = TR::VEC, ESZ = TR::ESZ;
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [c][2] moments, then [256][2*VEC] partials
    __shared__ float red_da[4];
    const int img = blockIdx.y;
    signal_start(p.start_flag, p.start_val);
    float* part = sm + 2 * p.c;
    load_moments(p.stats, img, p.c, p.npix, p.eps, sm);
    __syncthreads();
    const NormLane l = norm_lane<VEC>(p);
    const int cv_n = p.c / VEC;
    const float alpha = *p.alpha;
    float s12[2 * VEC], da = 0.f;
    float* const s1 = s12; float* const s2 = s12 + VEC;
#pragma unroll
    for (int e = 0; e < 2 * VEC; ++e) s12[e] = 0.f;
    if (l.active) {
        const DropKeys dk = dropout_keys(p.seed, p.layer, img, p.drop_thresh);
        const unsigned cq = p.c / 4, gq = l.ch0 / 4;
        const float scale = p.drop_scale;
        float mean[VEC], rstd[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) { mean[e] = sm[(l.ch0 + e) * 2]; rstd[e] = sm[(l.ch0 + e) * 2 + 1]; }
        const size_t img_pix = (size_t)img * p.npix;
        const char* xb = p.x + (img_pix * p.x_pitch + p.x_coff) * ESZ;
        const char* gb = p.dy + (img_pix * p.dy_pitch + p.dy_coff) * ESZ;
        const unsigned xs = p.x_pitch * ESZ, gs = p.dy_pitch * ESZ, cb = l.ch0 * ESZ;
        auto unit = [&](const uint4& xr, const uint4& gr, int pix) {
            float x[VEC], dy[VEC];
            bool keep[VEC];
            unpack_vec<TR>(xr, x); unpack_vec<TR>(gr, dy);
            if (DROP) {
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) dropout_keep4(dk, (unsigned)pix * cq + gq + q, keep + 4 * q);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xh = (x[e] - mean[e]) * rstd[e];
                float d = xh, sc = 1.f;
                if (DROP) { sc = keep[e] ? scale : 0.f; d = xh * sc; }
                const float gd = dy[e] * (d > 0.f ? 1.f : alpha);
                da += dy[e] * fminf(d, 0.f);
                const float gh = DROP ? gd * sc : gd;
                s1[e] += gh; s2[e] += gh * xh;
            }
        };
        int pix0 = l.pbeg + l.prow;
        for (; pix0 + (NORM_UB - 1) * l.rows < l.pend; pix0 += l.rows * NORM_UB) {
            uint4 xr[NORM_UB], gr[NORM_UB];
#pragma unroll
            for (int u = 0; u < NORM_UB; ++u) {
                const int pix = pix0 + u * l.rows;
                xr[u] = *reinterpret_cast<const uint4*>(xb + ((unsigned)pix * xs + cb));
                gr[u] = *reinterpret_cast<const uint4*>(gb + ((unsigned)pix * gs + cb));
            }
#pragma unroll
            for (int u = 0; u < NORM_UB; ++u) { unit(xr[u], gr[u], pix0 + u * l.rows); __builtin_amdgcn_sched_barrier(0); }
        }
        for (; pix0 < l.pend; pix0 += l.rows)
            unit(*reinterpret_cast<const uint4*>(xb + ((unsigned)pix0 * xs + cb)), *reinterpret_cast<const uint4*>(gb + ((unsigned)pix0 * gs + cb)), pix0);
    }
    da = wave_sum(da);
    if ((threadIdx.x & 63) == 0) red_da[threadIdx.x >> 6] = da;
    sum_rows<VEC>(s12, part, l, cv_n);
    const int slot = blockIdx.x % MI_NSLOT;
    for (int i = threadIdx.x; i < 2 * p.c; i += 256) {
        const int ch = i >> 1, k = i & 1;
        const int ccv = ch / VEC, e = ch - ccv * VEC;
        atomicAdd(&p.bstats[(((size_t)img * MI_NSLOT + slot) * p.c) * 2 + i], (double)part[ccv * 2 * VEC + k * VEC + e]);
    }
    if (threadIdx.x == 0 && p.dalpha) {
        const float v = ((red_da[0] + red_da[1]) + red_da[2]) + red_da[3];
        if (p.dalpha_slots) p.dalpha[img * gridDim.x + blockIdx.x] = v;
        else atomicAdd(p.dalpha, v);
    }
}

// backward pass 2: dx
template <class TR, bool DROP>
__global__ __launch_bounds__(256) void norm_bwd_apply_kernel(const NormParams p) {
    constexpr int VEC = TR::VEC, ESZ = TR::ESZ;
    extern __shared__ __attribute__((aligned(16))) float sm[];  // [c][2] moments, [c][2] backward means
    const int img = blockIdx.y;
    float* bm = sm + 2 * p.c;
    load_moments(p.stats, img, p.c, p.npix, p.eps, sm);
    for (int i = threadIdx.x; i < 2 * p.c; i += 256) {
        double a = 0.0;
#pragma unroll
        for (int sl = 0; sl < MI_NSLOT; ++sl) a += p.bstats[(((size_t)img * MI_NSLOT + sl) * p.c) * 2 + i];
        bm[i] = (float)(a / (double)p.npix);
    }
    __syncthreads();
    const NormLane l = norm_lane<VEC>(p);
    if (!l.active) return;
    const float alpha = *p.alpha;
    const DropKeys dk = dropout_keys(p.seed, p.layer, img, p.drop_thresh);
    const unsigned cq = p.c / 4, gq = l.ch0 / 4;
    const float scale = p.drop_scale;
    float mean[VEC], rstd[VEC], b1[VEC], b2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        mean[e] = sm[(l.ch0 + e) * 2]; rstd[e] = sm[(l.ch0 + e) * 2 + 1];
        b1[e] = bm[(l.ch0 + e) * 2]; b2[e] = bm[(l.ch0 + e) * 2 + 1];
    }
    const size_t img_pix = (size_t)img * p.npix;
    const char* xb = p.x + (img_pix * p.x_pitch + p.x_coff) * ESZ;
    const char* gb = p.dy + (img_pix * p.dy_pitch + p.dy_coff) * ESZ;
    char* ob = p.out + (img_pix * p.out_pitch + p.out_coff) * ESZ;
    const unsigned xs = p.x_pitch * ESZ, gs = p.dy_pitch * ESZ, os = p.out_pitch * ESZ, cb = l.ch0 * ESZ;
    auto unit = [&](const uint4& xr, const uint4& gr, int pix) {
        float x[VEC], dy[VEC], o[VEC];
        bool keep[VEC];
        unpack_vec<TR>(xr, x); unpack_vec<TR>(gr, dy);
        if (DROP) {
#pragma unroll
            for (int q = 0; q < VEC / 4; ++q) dropout_keep4(dk, (unsigned)pix * cq + gq + q, keep + 4 * q);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float xh = (x[e] - mean[e]) * rstd[e];
            float d = xh, sc = 1.f;
            if (DROP) { sc = keep[e] ? scale : 0.f; d = xh * sc; }
            float gh = dy[e] * (d > 0.f ? 1.f : alpha);
            if (DROP) gh *= sc;
            o[e] = rstd[e] * (gh - b1[e] - xh * b2[e]);
        }
        store_vec<TR>(ob + ((unsigned)pix * os + cb), o);
    };
    int pix0 = l.pbeg + l.prow;
    for (; pix0 + (NORM_UB - 1) * l.rows < l.pend; pix0 += l.rows * NORM_UB) {
        uint4 xr[NORM_UB], gr[NORM_UB];
#pragma unroll
        for (int u = 0; u < NORM_UB; ++u) {
            const int pix = pix0 + u * l.rows;
            xr[u] = *reinterpret_cast<const uint4*>(xb + ((unsigned)pix * xs + cb));
            gr[u] = *reinterpret_cast<const uint4*>(gb + ((unsigned)pix * gs + cb));
        }
#pragma unroll
        for (int u = 0; u < NORM_UB; ++u) { unit(xr[u], gr[u], pix0 + u * l.rows); __builtin_amdgcn_sched_barrier(0); }
    }
    for (; pix0 < l.pend; pix0 += l.rows)
        unit(*reinterpret_cast<const uint4*>(xb + ((unsigned)pix0 * xs + cb)), *reinterpret_cast<const uint4*>(gb + ((unsigned)pix0 * gs + cb)), pix0);
}

// ---- small images: one workgroup per (image, eight channels) -----------------------------------------------------------
// For tensors of a few hundred kilobytes the streaming kernels are dependent launches of a few microseconds each with
// nothing to do.  The kernels below give a workgroup of T threads one image and SM_CG channels; a thread keeps its
// pixels (at most SmKeep sweeps of the workgroup over the image) in registers, so every reduction over the image stays
// inside the workgroup: no global accumulators, no second pass over memory.  At the sizes of a large batch they lose
// (32 bytes per pixel and workgroup is a poor access shape), so the launchers take them only for small tensors.
constexpr int SM_CG = 8;
template <class TR> struct SmKeep { static constexpr int K = TR::ESZ == 2 ? 2 : 4; };

// v[VEC] = sum over the shares of a contraction-split convolution, in share order.  The shares sit `stride` floats
// apart; all KS loads are issued before the first use, shares past the last one (index ksl) re-read it and add nothing.
template <int VEC, int KS>
__device__ __forceinline__ void sum_shares(const float* q, size_t stride, int ksl, float* v) {
    float4 t[KS][VEC / 4];
#pragma unroll
    for (int ks = 0; ks < KS; ++ks)
#pragma unroll
        for (int g = 0; g < VEC / 4; ++g)
            t[ks][g] = *reinterpret_cast<const float4*>(q + (ks < ksl ? ks : ksl) * stride + 4 * g);
#pragma unroll
    for (int e = 0; e < VEC; ++e) v[e] = 0.f;
#pragma unroll
    for (int ks = 0; ks < KS; ++ks) {
        const float m = ks <= ksl ? 1.f : 0.f;
#pragma unroll
        for (int g = 0; g < VEC / 4; ++g) {
            v[4 * g] += m * t[ks][g].x; v[4 * g + 1] += m * t[ks][g].y;
            v[4 * g + 2] += m * t[ks][g].z; v[4 * g + 3] += m * t[ks][g].w;
        }
    }
}
template <int VEC>
__device__ __forceinline__ void sum_shares_any(const float* q, size_t stride, int ksl, float* v) {
    if (ksl < 2) sum_shares<VEC, 2>(q, stride, ksl, v);
    else if (ksl < 4) sum_shares<VEC, 4>(q, stride, ksl, v);
    else sum_shares<VEC, 8>(q, stride, ksl, v);
}

// Folds s[W] over the pixel rows of the workgroup (lanes of one channel vector sit CVW apart inside a wave, the waves
// meet in LDS in wave order); after the barrier part[0][cv][e] holds the totals.
template <int W, int CVW, int NW>
__device__ __forceinline__ void fold_rows(float* s, float (&part)[NW][CVW][W]) {
#pragma unroll
    for (int e = 0; e < W; ++e) {
        float t = s[e];
#pragma unroll
        for (int o = 32; o >= CVW; o >>= 1) t += __shfl_xor(t, o, 64);
        s[e] = t;
    }
    const int lane = threadIdx.x & 63, w = threadIdx.x >> 6;
    if (lane < CVW) {
#pragma unroll
        for (int e = 0; e < W; ++e) part[w][lane][e] = s[e];
    }
    __syncthreads();
    if (threadIdx.x < CVW * W) {
        const int vcv = threadIdx.x / W, ve = threadIdx.x % W;
        float t = part[0][vcv][ve];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) t += part[ww][vcv][ve];
        part[0][vcv][ve] = t;
    }
    __syncthreads();
}

// Norm backward in ONE launch.  PARTS: dy is not read from a tensor but completed here from the partial sums of the
// contraction-split data-gradient convolution in front of this norm (shares [split][image][N / 8][pixel][8], plus its
// addend), rounded and stored to that convolution's destination on the way: the finishing pass of that layer.
// Same arithmetic per element as the two streaming kernels.
template <class TR, bool DROP, int T, bool PARTS>
__global__ __launch_bounds__(T) void norm_bwd_small_kernel(const NormParams p, const IgemmParams ip) {
    constexpr int VEC = TR::VEC, ESZ = TR::ESZ;
    constexpr int CVW = SM_CG / VEC, ROWS = T / CVW, NW = T / 64, KEEP = SmKeep<TR>::K;
    __shared__ float mom[SM_CG * 2];
    __shared__ float part[NW][CVW][2 * VEC];
    __shared__ float red_da[NW];
    const int img = blockIdx.y;
    const int c0 = blockIdx.x * SM_CG;
    const int npix = p.npix;
    signal_start(p.start_flag, p.start_val);
    if (threadIdx.x < SM_CG) {
        const int ch = c0 + threadIdx.x;
        double s1 = 0.0, s2 = 0.0;
#pragma unroll
        for (int sl = 0; sl < MI_NSLOT; ++sl) {
            const double* q = p.stats + (((size_t)img * MI_NSLOT + sl) * p.c + ch) * 2;
            s1 += q[0]; s2 += q[1];
        }
        const double mean = s1 / (double)npix;
        double var = s2 / (double)npix - mean * mean;
        var = var > 0.0 ? var : 0.0;
        mom[threadIdx.x * 2] = (float)mean;
        mom[threadIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    const int cv = threadIdx.x % CVW, prow = threadIdx.x / CVW;
    const int ch0 = c0 + cv * VEC;
    const size_t img_pix = (size_t)img * npix;
    const char* xb = p.x + (img_pix * p.x_pitch + p.x_coff + ch0) * ESZ;
    char* ob = p.out + (img_pix * p.out_pitch + p.out_coff + ch0) * ESZ;
    const unsigned xs = p.x_pitch * ESZ, os = p.out_pitch * ESZ;
    uint4 xr[KEEP], gr[KEEP];
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const int pix = prow + k * ROWS;
        xr[k] = *reinterpret_cast<const uint4*>(xb + (unsigned)(pix < npix ? pix : npix - 1) * xs);
    }
    if (PARTS) {
        const float* kp = ip.kpart + ((size_t)img * (ip.N / SM_CG) + blockIdx.x) * (size_t)npix * SM_CG + cv * VEC;
        const size_t stride = (size_t)ip.nb * npix * ip.N;
        const int ksl = ip.ksplit - 1;
        const char* ab = ip.addend ? ip.addend + (img_pix * ip.add_pitch + ip.add_coff + ch0) * ESZ : nullptr;
        char* db = ip.out + (img_pix * ip.out_pitch + ip.out_coff + ch0) * ESZ;
        const unsigned as = ip.add_pitch * ESZ, ds = ip.out_pitch * ESZ;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int pix = prow + k * ROWS;
            const int pc = pix < npix ? pix : npix - 1;
            uint4 ar = make_uint4(0u, 0u, 0u, 0u);
            if (ab) ar = *reinterpret_cast<const uint4*>(ab + (unsigned)pc * as);
            float v[VEC];
            sum_shares_any<VEC>(kp + (size_t)pc * SM_CG, stride, ksl, v);
            if (ab) {
                float a[VEC];
                unpack_vec<TR>(ar, a);
#pragma unroll
                for (int e = 0; e < VEC; ++e) v[e] += a[e];
            }
            gr[k] = pack_vec<TR>(v);
            if (pix < npix) *reinterpret_cast<uint4*>(db + (unsigned)pix * ds) = gr[k];
            __builtin_amdgcn_sched_barrier(0);
        }
    } else {
        const char* gb = p.dy + (img_pix * p.dy_pitch + p.dy_coff + ch0) * ESZ;
        const unsigned gs = p.dy_pitch * ESZ;
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int pix = prow + k * ROWS;
            gr[k] = *reinterpret_cast<const uint4*>(gb + (unsigned)(pix < npix ? pix : npix - 1) * gs);
        }
    }
    __syncthreads();
    const float alpha = *p.alpha;
    const DropKeys dk = dropout_keys(p.seed, p.layer, img, p.drop_thresh);
    const unsigned cq = p.c / 4, gq = ch0 / 4;
    const float scale = p.drop_scale;
    float mean[VEC], rstd[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mean[e] = mom[(cv * VEC + e) * 2]; rstd[e] = mom[(cv * VEC + e) * 2 + 1]; }
    float s12[2 * VEC], da = 0.f;
#pragma unroll
    for (int e = 0; e < 2 * VEC; ++e) s12[e] = 0.f;
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const int pix = prow + k * ROWS;
        if (pix < npix) {
            float x[VEC], dy[VEC];
            bool keep[VEC];
            unpack_vec<TR>(xr[k], x); unpack_vec<TR>(gr[k], dy);
            if (DROP) {
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) dropout_keep4(dk, (unsigned)pix * cq + gq + q, keep + 4 * q);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xh = (x[e] - mean[e]) * rstd[e];
                float d = xh, sc = 1.f;
                if (DROP) { sc = keep[e] ? scale : 0.f; d = xh * sc; }
                const float gd = dy[e] * (d > 0.f ? 1.f : alpha);
                da += dy[e] * fminf(d, 0.f);
                const float gh = DROP ? gd * sc : gd;
                s12[e] += gh; s12[VEC + e] += gh * xh;
            }
        }
    }
    da = wave_sum(da);
    if ((threadIdx.x & 63) == 0) red_da[threadIdx.x >> 6] = da;
    fold_rows<2 * VEC, CVW, NW>(s12, part);       // its barriers make red_da visible too
    if (threadIdx.x == 0 && p.dalpha) {
        float v = red_da[0];
#pragma unroll
        for (int ww = 1; ww < NW; ++ww) v += red_da[ww];
        if (p.dalpha_slots) p.dalpha[img * gridDim.x + blockIdx.x] = v;
        else atomicAdd(p.dalpha, v);
    }
    const float inv_n = 1.0f / (float)npix;
    float b1[VEC], b2[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { b1[e] = part[0][cv][e] * inv_n; b2[e] = part[0][cv][VEC + e] * inv_n; }
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const int pix = prow + k * ROWS;
        if (pix < npix) {
            float x[VEC], dy[VEC], o[VEC];
            bool keep[VEC];
            unpack_vec<TR>(xr[k], x); unpack_vec<TR>(gr[k], dy);
            if (DROP) {
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) dropout_keep4(dk, (unsigned)pix * cq + gq + q, keep + 4 * q);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                const float xh = (x[e] - mean[e]) * rstd[e];
                float d = xh, sc = 1.f;
                if (DROP) { sc = keep[e] ? scale : 0.f; d = xh * sc; }
                float gh = dy[e] * (d > 0.f ? 1.f : alpha);
                if (DROP) gh *= sc;
                o[e] = rstd[e] * (gh - b1[e] - xh * b2[e]);
            }
            store_vec<TR>(ob + (unsigned)pix * os, o);
        }
    }
}

// Contraction-split layers of small images: the finishing pass of the convolution and the norm forward in ONE
// launch.  A workgroup of T threads owns one image and SM_CG output channels of the convolution; a thread keeps its
// pixels (at most FN_KEEP sweeps of the workgroup over the image) in registers:
//   pass 1: v[pixel][ch] = sum_split partial[split][pixel][ch] (+ bias), rounded to the activation type and stored to
//           the convolution's destination(s); the per-channel sums of the values and their squares on the way;
//   then (normalised channels only) the workgroup folds its sums, writes them to slot 0 of the forward statistics
//   (the backward pass reads them there) and takes mean / rstd from them;
//   pass 2: norm, dropout, PReLU and the residual on the kept (rounded) values, exactly as norm_fwd_kernel does.
// Channel groups past the normalised channels (the residual half of a fused two-output convolution) stop after pass 1.
template <class TR, bool DROP, bool RES, int T>
__global__ __launch_bounds__(T) void finish_norm_kernel(const IgemmParams ip, const NormParams p) {
    constexpr int VEC = TR::VEC, ESZ = TR::ESZ;
    constexpr int CVW = SM_CG / VEC, ROWS = T / CVW, NW = T / 64, KEEP = SmKeep<TR>::K;
    __shared__ float part[NW][CVW][2 * VEC];
    __shared__ float mom[SM_CG * 2];
    const int img = blockIdx.y, c0 = blockIdx.x * SM_CG;
    const int cv = threadIdx.x % CVW, prow = threadIdx.x / CVW;
    const int ch = c0 + cv * VEC;
    const int npix = p.npix;
    const size_t npo = (size_t)ip.nb * npix, img_pix = (size_t)img * npix;
    const bool normed = c0 < ip.nstat;                       // uniform over the workgroup
    const bool first = ch < ip.out0_c;
    char* const rawb = first ? ip.out + (img_pix * ip.out_pitch + ip.out_coff + ch) * ESZ
                             : ip.out1 + (img_pix * ip.out1_pitch + ip.out1_coff + ch - ip.out0_c) * ESZ;
    const unsigned raws = (first ? ip.out_pitch : ip.out1_pitch) * ESZ;
    float bias[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) bias[e] = (ip.bias && ch >= ip.bias_c0) ? ip.bias[ch - ip.bias_c0 + e] : 0.f;
    float s12[2 * VEC];
#pragma unroll
    for (int e = 0; e < 2 * VEC; ++e) s12[e] = 0.f;
    // partial sums [split][image][N / 8][pixel][8]: one contiguous run per (split, image, channel group)
    const float* kp = ip.kpart + ((size_t)img * (ip.N / SM_CG) + blockIdx.x) * (size_t)npix * SM_CG + cv * VEC;
    const size_t ksplit_stride = npo * ip.N;
    const int ksl = ip.ksplit - 1;
    uint4 kept[KEEP];
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const int pix = prow + k * ROWS;
        const bool valid = pix < npix;
        const float* q = kp + (size_t)(valid ? pix : npix - 1) * SM_CG;
        float v[VEC];
        sum_shares_any<VEC>(q, ksplit_stride, ksl, v);
        if (valid) {
#pragma unroll
            for (int e = 0; e < VEC; ++e) { s12[e] += v[e]; s12[VEC + e] += v[e] * v[e]; v[e] += bias[e]; }
            kept[k] = pack_vec<TR>(v);
            *reinterpret_cast<uint4*>(rawb + (unsigned)pix * raws) = kept[k];
        }
        __builtin_amdgcn_sched_barrier(0);
    }
    if (!normed) return;
    fold_rows<2 * VEC, CVW, NW>(s12, part);
    if (threadIdx.x < SM_CG) {
        const int vcv = threadIdx.x / VEC, ve = threadIdx.x % VEC;
        const float t1 = part[0][vcv][ve], t2 = part[0][vcv][VEC + ve];
        double* sq = ip.stats + (((size_t)img * MI_NSLOT) * ip.stats_c + c0 + threadIdx.x) * 2;
        sq[0] = (double)t1; sq[1] = (double)t2;
        const double mean = (double)t1 / (double)npix;
        double var = (double)t2 / (double)npix - mean * mean;
        var = var > 0.0 ? var : 0.0;
        mom[threadIdx.x * 2] = (float)mean;
        mom[threadIdx.x * 2 + 1] = (float)(1.0 / sqrt(var + (double)p.eps));
    }
    const char* rb = RES ? p.res + (img_pix * p.res_pitch + p.res_coff + ch) * ESZ : nullptr;
    const unsigned rs = p.res_pitch * ESZ, os = p.out_pitch * ESZ;
    uint4 rr[KEEP];
    if (RES) {
#pragma unroll
        for (int k = 0; k < KEEP; ++k) {
            const int pix = prow + k * ROWS;
            rr[k] = *reinterpret_cast<const uint4*>(rb + (unsigned)(pix < npix ? pix : npix - 1) * rs);
        }
    }
    __syncthreads();
    const float alpha = *p.alpha;
    const DropKeys dk = dropout_keys(p.seed, p.layer, img, p.drop_thresh);
    const unsigned cq = p.c / 4, gq = ch / 4;
    const float scale = p.drop_scale;
    float mean[VEC], rstd[VEC];
#pragma unroll
    for (int e = 0; e < VEC; ++e) { mean[e] = mom[(cv * VEC + e) * 2]; rstd[e] = mom[(cv * VEC + e) * 2 + 1]; }
    char* ob = p.out + (img_pix * p.out_pitch + p.out_coff + ch) * ESZ;
#pragma unroll
    for (int k = 0; k < KEEP; ++k) {
        const int pix = prow + k * ROWS;
        if (pix < npix) {
            float x[VEC], r[VEC], o[VEC];
            bool keep[VEC];
            unpack_vec<TR>(kept[k], x);
            if (RES) unpack_vec<TR>(rr[k], r);
            if (DROP) {
#pragma unroll
                for (int q = 0; q < VEC / 4; ++q) dropout_keep4(dk, (unsigned)pix * cq + gq + q, keep + 4 * q);
            }
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                float d = (x[e] - mean[e]) * rstd[e];
                if (DROP) d = keep[e] ? d * scale : 0.f;
                float y = d > 0.f ? d : alpha * d;
                if (RES) y += r[e];
                o[e] = y;
            }
            store_vec<TR>(ob + (unsigned)pix * os, o);
        }
    }
}

// ---- one-channel fp32 planar variants (top of the decoder) ------------------------
// tensors are [nb][npix] fp32; four consecutive pixels per lane.
// W = 4: four consecutive pixels per lane as one 16-byte access (npix a multiple of 4; they are exactly one
// dropout group); W = 1: any npix.
template <int W> struct PlanarVec;
template <> struct PlanarVec<4> { using T = float4; };
template <> struct PlanarVec<1> { using T = float; };
template <int W>
__device__ __forceinline__ void planar_load(const float* p, int g, float* v) {
    const typename PlanarVec<W>::T t = reinterpret_cast<const typename PlanarVec<W>::T*>(p)[g];
    if constexpr (W == 4) { v[0] = t.x; v[1] = t.y; v[2] = t.z; v[3] = t.w; } else v[0] = t;
}
template <int W>
__device__ __forceinline__ void planar_store(float* p, int g, const float* v) {
    if constexpr (W == 4) reinterpret_cast<float4*>(p)[g] = make_float4(v[0], v[1], v[2], v[3]);
    else p[g] = v[0];
}
template <bool DROP, int W>
__device__ __forceinline__ void planar_keep(const DropKeys& dk, int g, bool* keep) {
    if (DROP) {
        if constexpr (W == 4) dropout_keep4(dk, (unsigned)g, keep);
        else keep[0] = dropout_keep1(dk, (unsigned)g);
    }
}

template <bool DROP, int W>
__global__ __launch_bounds__(256) void norm1_fwd_kernel(const NormParams p) {
    const int img = blockIdx.y;
    float mean, rstd;
    c1_moments(p.stats, img, p.npix, p.eps, mean, rstd);
    const float alpha = *p.alpha;
    const DropKeys dk = dropout_keys(p.seed, p.layer, img, p.drop_thresh);
    const float scale = p.drop_scale;
    const float* x = reinterpret_cast<const float*>(p.x) + (size_t)img * p.npix;
    float* out = reinterpret_cast<float*>(p.out) + (size_t)img * p.npix;
    const int ng = p.npix / W;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) {
        float v[W], o[W];
        bool keep[W];
        planar_load<W>(x, g, v);
        planar_keep<DROP, W>(dk, g, keep);
#pragma unroll
        for (int j = 0; j < W; ++j) {
            float d = (v[j] - mean) * rstd;
            if (DROP) d = keep[j] ? d * scale : 0.f;
            o[j] = d > 0.f ? d : alpha * d;
        }
        planar_store<W>(out, g, o);
    }
}

template <bool DROP, int W>
__global__ __launch_bounds__(256) void norm1_bwd_reduce_kernel(const NormParams p) {
    __shared__ float red[3][4];
    const int img = blockIdx.y;
    signal_start(p.start_flag, p.start_val);
    float mean, rstd;
    c1_moments(p.stats, img, p.npix, p.eps, mean, rstd);
    const float alpha = *p.alpha;
    const DropKeys dk = dropout_keys(p.seed, p.layer, img, p.drop_thresh);
    const float scale = p.drop_scale;
    const float* x = reinterpret_cast<const float*>(p.x) + (size_t)img * p.npix;
    const float* dy = reinterpret_cast<const float*>(p.dy) + (size_t)img * p.npix;
    float s1 = 0.f, s2 = 0.f, da = 0.f;
    const int ng = p.npix / W;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) {
        float v[W], gy[W];
        bool keep[W];
        planar_load<W>(x, g, v);
        planar_load<W>(dy, g, gy);
        planar_keep<DROP, W>(dk, g, keep);
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const float xh = (v[j] - mean) * rstd;
            float d = xh, sc = 1.f;
            if (DROP) { sc = keep[j] ? scale : 0.f; d = xh * sc; }
            float gh = gy[j] * (d > 0.f ? 1.f : alpha);
            if (DROP) gh *= sc;
            da += gy[j] * fminf(d, 0.f);
            s1 += gh; s2 += gh * xh;
        }
    }
    s1 = wave_sum(s1); s2 = wave_sum(s2); da = wave_sum(da);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = s1; red[1][w] = s2; red[2][w] = da; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const int slot = blockIdx.x % MI_NSLOT;
        atomicAdd(&p.bstats[((size_t)img * MI_NSLOT + slot) * 2], (double)(((red[0][0] + red[0][1]) + red[0][2]) + red[0][3]));
        atomicAdd(&p.bstats[((size_t)img * MI_NSLOT + slot) * 2 + 1], (double)(((red[1][0] + red[1][1]) + red[1][2]) + red[1][3]));
        if (p.dalpha) {
            const float v = ((red[2][0] + red[2][1]) + red[2][2]) + red[2][3];
            if (p.dalpha_slots) p.dalpha[img * gridDim.x + blockIdx.x] = v;
            else atomicAdd(p.dalpha, v);
        }
    }
}

template <bool DROP, int W>
__global__ __launch_bounds__(256) void norm1_bwd_apply_kernel(const NormParams p) {
    const int img = blockIdx.y;
    float mean, rstd;
    c1_moments(p.stats, img, p.npix, p.eps, mean, rstd);
    double d1 = 0.0, d2 = 0.0;
#pragma unroll
    for (int sl = 0; sl < MI_NSLOT; ++sl) {
        d1 += p.bstats[((size_t)img * MI_NSLOT + sl) * 2];
        d2 += p.bstats[((size_t)img * MI_NSLOT + sl) * 2 + 1];
    }
    const float b1 = (float)(d1 / (double)p.npix), b2 = (float)(d2 / (double)p.npix);
    const float alpha = *p.alpha;
    const DropKeys dk = dropout_keys(p.seed, p.layer, img, p.drop_thresh);
    const float scale = p.drop_scale;
    const float* x = reinterpret_cast<const float*>(p.x) + (size_t)img * p.npix;
    const float* dy = reinterpret_cast<const float*>(p.dy) + (size_t)img * p.npix;
    float* out = reinterpret_cast<float*>(p.out) + (size_t)img * p.npix;
    const int ng = p.npix / W;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) {
        float v[W], gy[W], o[W];
        bool keep[W];
        planar_load<W>(x, g, v);
        planar_load<W>(dy, g, gy);
        planar_keep<DROP, W>(dk, g, keep);
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const float xh = (v[j] - mean) * rstd;
            float d = xh, sc = 1.f;
            if (DROP) { sc = keep[j] ? scale : 0.f; d = xh * sc; }
            float gh = gy[j] * (d > 0.f ? 1.f : alpha);
            if (DROP) gh *= sc;
            o[j] = rstd * (gh - b1 - xh * b2);
        }
        planar_store<W>(out, g, o);
    }
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string(what) + ": " + hipGetErrorString(e)); return -2; }
    return 0;
}

}  // namespace

// workgroups per image of the streaming kernels: `sweeps` pixel sweeps per thread (a sweep of the workgroup
// covers 256 / (16-byte vectors per pixel) pixels)
static int norm_blocks(int npix, int c, int vec, int sweeps, int cap) {
    const int cv_n = c / vec;
    const int rows = 256 / cv_n;
    const int per_block = rows * sweeps;
    int g = (npix + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return g;
}

// The one-launch backward serves small tensors only: images of at most 256 / (vectors per pixel of a 16-channel group)
// x 16 pixels (bf16 2048, fp32 1024) and at most `max_elems` elements in the whole tensor (MI_NORM_BWD_SMALL_MAX, default
// 2^21: at the sizes of a large batch the two streaming kernels are faster; 0 switches the kernel off).
static bool norm_bwd_small_ok(int dtype, const NormParams& p) {
    static const long long max_elems = getenv("MI_NORM_BWD_SMALL_MAX") ? atoll(getenv("MI_NORM_BWD_SMALL_MAX")) : (1ll << 21);
    (void)dtype;
    return p.c % SM_CG == 0 && p.npix <= 2048 && (long long)p.nb * p.npix * p.c <= max_elems;
}

// workgroups per image of the backward reduction pass
static int norm_bwd_reduce_grid(int npix, int c) {
    if (c == 1) {
        int g = (npix + 2047) / 2048;
        return g < 1 ? 1 : g > 128 ? 128 : g;
    }
    // one count for both dtypes (the caller sizes the slope-gradient slots from it): 32 sweeps of an fp32
    // workgroup (16 of a bf16 one), but at least 16 workgroups per image while 4 sweeps remain
    const int g = norm_blocks(npix, c, 4, 32, 256);
    const int gmin = norm_blocks(npix, c, 4, 4, 16);
    return g > gmin ? g : gmin;
}

// slope-gradient partials per image a caller has to provide (zeroed: whichever kernel runs writes its workgroups'
// partials to the first slots and leaves the rest alone): the larger of the two kernels' workgroup counts
int norm_bwd_reduce_blocks(int npix, int c) {
    const int g = norm_bwd_reduce_grid(npix, c);
    const int gs = c % SM_CG == 0 ? c / SM_CG : 0;
    return g > gs ? g : gs;
}

// planar kernels: 16-byte accesses when every image starts on a 16-byte boundary
static bool planar_v4(const NormParams& p) {
    auto al = [](const void* q) { return ((uintptr_t)q & 15) == 0; };
    return p.npix % 4 == 0 && al(p.x) && al(p.out) && (!p.dy || al(p.dy));
}
#define NORM1_CASE(K, GRID)                                                                      \
    do {                                                                                         \
        if (p.drop_thresh) { if (v4) hipLaunchKernelGGL((K<true, 4>), GRID, dim3(256), 0, st, p); \
                             else hipLaunchKernelGGL((K<true, 1>), GRID, dim3(256), 0, st, p); }  \
        else { if (v4) hipLaunchKernelGGL((K<false, 4>), GRID, dim3(256), 0, st, p);              \
               else hipLaunchKernelGGL((K<false, 1>), GRID, dim3(256), 0, st, p); }               \
    } while (0)

// dtype: 0 fp32, 1 bf16.  c == 1 selects the planar fp32 kernels (pitch must be 1, dtype 0).
int launch_norm_fwd(int dtype, const NormParams& p, hipStream_t st) {
    {
        const long long pm = p.res ? (p.res_pitch > p.x_pitch ? p.res_pitch : p.x_pitch) : p.x_pitch;
        MI_CHECK((long long)p.npix * (pm > p.out_pitch ? pm : p.out_pitch) * (dtype ? 2 : 4) < (1ll << 32), "one image of a tensor must stay below 4 GiB");
    }
    MI_CHECK(!p.drop_thresh || (long long)p.npix * p.c < (1ll << 34), "dropout numbers the elements of one image in 34 bits");
    if (p.c == 1) {
        MI_CHECK(dtype == 0 && p.x_pitch == 1 && p.out_pitch == 1 && !p.res, "one-channel tensors are planar fp32");
        const bool v4 = planar_v4(p);
        int gx = (p.npix / (v4 ? 4 : 1) + 511) / 512; if (gx > 256) gx = 256; if (gx < 1) gx = 1;
        NORM1_CASE(norm1_fwd_kernel, dim3(gx, p.nb));
        return check_launch("norm1_fwd");
    }
    const int vec = dtype ? 8 : 4;
    MI_CHECK(p.c % vec == 0, "channels must be a multiple of the 16-byte vector");
    MI_CHECK(p.c / vec <= 256, "too many channels for the norm kernels");
    int gx = norm_blocks(p.npix, p.c, vec, 8, 65535);
    if ((long long)gx * p.nb < 1024) gx = norm_blocks(p.npix, p.c, vec, 4, 65535);   // small tensors: shorter workgroups, more of them
    const size_t lds = (size_t)p.c * 2 * sizeof(float);
    const dim3 grid(gx, p.nb);
#define NORM_FWD_CASE(TR, D, R) hipLaunchKernelGGL((norm_fwd_kernel<TR, D, R>), grid, dim3(256), lds, st, p)
    const bool drop = p.drop_thresh != 0, res = p.res != nullptr;
    if (dtype) {
        if (drop) { if (res) NORM_FWD_CASE(BF16, true, true); else NORM_FWD_CASE(BF16, true, false); }
        else { if (res) NORM_FWD_CASE(BF16, false, true); else NORM_FWD_CASE(BF16, false, false); }
    } else {
        if (drop) { if (res) NORM_FWD_CASE(F32, true, true); else NORM_FWD_CASE(F32, true, false); }
        else { if (res) NORM_FWD_CASE(F32, false, true); else NORM_FWD_CASE(F32, false, false); }
    }
#undef NORM_FWD_CASE
    return check_launch("norm_fwd");
}

// Finishing pass of a contraction-split convolution and the norm forward behind it as one launch: small images only
// (a workgroup walks a whole image), convolutions without an addend whose statistics cover exactly the normalised
// channels.  MI_NO_FINISH_NORM=1 keeps the two launches.
bool finish_norm_ok(const IgemmParams& ip, int c, int npix) {
    static const bool off = getenv("MI_NO_FINISH_NORM") != nullptr;
    if (off || ip.ksplit < 2 || ip.ksplit > 8 || ip.addend || !ip.stats) return false;
    if (ip.N % SM_CG || c % SM_CG || ip.nstat != c || ip.stats_c != c) return false;
    if (ip.out1 && (ip.out0_c % SM_CG || ip.out0_c != c)) return false;
    if (!ip.out1 && ip.N != c) return false;
    return npix <= 2048 && npix == ip.od[0] * ip.od[1] * ip.od[2];      // FN_KEEP sweeps of 1024 threads
}

int launch_finish_norm(int dtype, const IgemmParams& pin, const NormParams& p, hipStream_t st) {
    IgemmParams ip = pin;
    if (!ip.out1) ip.out0_c = ip.N;
    MI_CHECK(finish_norm_ok(pin, p.c, p.npix) && pin.kpart_blocked, "finish + norm launch on a layer it does not serve");
    {
        long long pm = ip.out_pitch > p.out_pitch ? ip.out_pitch : p.out_pitch;
        if (ip.out1 && ip.out1_pitch > pm) pm = ip.out1_pitch;
        if (p.res && p.res_pitch > pm) pm = p.res_pitch;
        MI_CHECK((long long)p.npix * pm * (dtype ? 2 : 4) < (1ll << 32), "one image of a tensor must stay below 4 GiB");
    }
    const dim3 grid(ip.N / SM_CG, ip.nb);
    const bool small = p.npix <= 512;       // FN_KEEP sweeps of 256 threads cover the image
#define FN_CASE(TR, D, R)                                                                                   \
    do {                                                                                                    \
        if (small) hipLaunchKernelGGL((finish_norm_kernel<TR, D, R, 256>), grid, dim3(256), 0, st, ip, p);   \
        else hipLaunchKernelGGL((finish_norm_kernel<TR, D, R, 1024>), grid, dim3(1024), 0, st, ip, p);       \
    } while (0)
    const bool drop = p.drop_thresh != 0, res = p.res != nullptr;
    if (dtype) {
        if (drop) { if (res) FN_CASE(BF16, true, true); else FN_CASE(BF16, true, false); }
        else { if (res) FN_CASE(BF16, false, true); else FN_CASE(BF16, false, false); }
    } else {
        if (drop) { if (res) FN_CASE(F32, true, true); else FN_CASE(F32, true, false); }
        else { if (res) FN_CASE(F32, false, true); else FN_CASE(F32, false, false); }
    }
#undef FN_CASE
    return check_launch("finish_norm");
}

// One-launch norm backward of a small tensor; with `ip` the gradient dy is completed here from the partial sums of the
// contraction-split convolution `ip` (see norm_bwd_small_kernel).
static int launch_norm_bwd_small(int dtype, const NormParams& p, const IgemmParams* ip, hipStream_t st) {
    IgemmParams q = IgemmParams();
    if (ip) q = *ip;
    const dim3 grid(p.c / SM_CG, p.nb);
    const bool drop = p.drop_thresh != 0, small = p.npix <= 512, parts = ip != nullptr;
#define NBS_CASE(TR, D)                                                                                                  \
    do {                                                                                                                 \
        if (small) { if (parts) hipLaunchKernelGGL((norm_bwd_small_kernel<TR, D, 256, true>), grid, dim3(256), 0, st, p, q);     \
                     else hipLaunchKernelGGL((norm_bwd_small_kernel<TR, D, 256, false>), grid, dim3(256), 0, st, p, q); }        \
        else { if (parts) hipLaunchKernelGGL((norm_bwd_small_kernel<TR, D, 1024, true>), grid, dim3(1024), 0, st, p, q);         \
               else hipLaunchKernelGGL((norm_bwd_small_kernel<TR, D, 1024, false>), grid, dim3(1024), 0, st, p, q); }            \
    } while (0)
    if (dtype) { if (drop) NBS_CASE(BF16, true); else NBS_CASE(BF16, false); }
    else { if (drop) NBS_CASE(F32, true); else NBS_CASE(F32, false); }
#undef NBS_CASE
    return check_launch(parts ? "finish_norm_bwd" : "norm_bwd_small");
}

// A contraction-split data-gradient convolution whose output is the dy of a small norm backward: one launch finishes
// the convolution (shares + addend -> dy, stored) and runs the norm backward.  MI_NO_FINISH_NORM=1 keeps two launches.
bool finish_norm_bwd_ok(const IgemmParams& ip, int dtype, int c, int npix, int nb) {
    static const bool off = getenv("MI_NO_FINISH_NORM") != nullptr;
    if (off || ip.ksplit < 2 || ip.ksplit > 8 || ip.stats || ip.bias || ip.out1) return false;
    if (ip.N != c || c % SM_CG || npix != ip.od[0] * ip.od[1] * ip.od[2] || nb != ip.nb) return false;
    NormParams t = NormParams(); t.c = c; t.npix = npix; t.nb = nb;
    return norm_bwd_small_ok(dtype, t);
}
int launch_finish_norm_bwd(int dtype, const IgemmParams& ip, const NormParams& p, hipStream_t st) {
    MI_CHECK(finish_norm_bwd_ok(ip, dtype, p.c, p.npix, p.nb) && ip.kpart_blocked, "finish + norm backward launch on a layer it does not serve");
    {
        long long pm = p.x_pitch > p.out_pitch ? p.x_pitch : p.out_pitch;
        if (ip.out_pitch > pm) pm = ip.out_pitch;
        if (ip.addend && ip.add_pitch > pm) pm = ip.add_pitch;
        MI_CHECK((long long)p.npix * pm * (dtype ? 2 : 4) < (1ll << 32), "one image of a tensor must stay below 4 GiB");
    }
    MI_CHECK(!p.drop_thresh || (long long)p.npix * p.c < (1ll << 34), "dropout numbers the elements of one image in 34 bits");
    return launch_norm_bwd_small(dtype, p, &ip, st);
}

// bstats ([nb][NSLOT][c][2] doubles) must be zero on entry; dalpha is accumulated into.
int launch_norm_bwd(int dtype, const NormParams& p, hipStream_t st) {
    {
        const long long pm = p.dy_pitch > p.x_pitch ? p.dy_pitch : p.x_pitch;
        MI_CHECK((long long)p.npix * (pm > p.out_pitch ? pm : p.out_pitch) * (dtype ? 2 : 4) < (1ll << 32), "one image of a tensor must stay below 4 GiB");
    }
    MI_CHECK(!p.drop_thresh || (long long)p.npix * p.c < (1ll << 34), "dropout numbers the elements of one image in 34 bits");
    if (p.c == 1) {
        MI_CHECK(dtype == 0 && p.x_pitch == 1 && p.out_pitch == 1 && p.dy_pitch == 1, "one-channel tensors are planar fp32");
        const bool v4 = planar_v4(p);
        const int gx = norm_bwd_reduce_grid(p.npix, 1);
        NORM1_CASE(norm1_bwd_reduce_kernel, dim3(gx, p.nb));
        if (int r = check_launch("norm1_bwd_reduce")) return r;
        int ga = (p.npix / (v4 ? 4 : 1) + 511) / 512; if (ga > 256) ga = 256; if (ga < 1) ga = 1;
        NORM1_CASE(norm1_bwd_apply_kernel, dim3(ga, p.nb));
        return check_launch("norm1_bwd_apply");
    }
    const int vec = dtype ? 8 : 4;
    MI_CHECK(p.c % vec == 0, "channels must be a multiple of the 16-byte vector");
    MI_CHECK(p.c / vec <= 256, "too many channels for the reduction kernel");
    const bool drop = p.drop_thresh != 0;
    if (norm_bwd_small_ok(dtype, p)) return launch_norm_bwd_small(dtype, p, nullptr, st);
    const size_t lds = (size_t)p.c * 4 * sizeof(float);
    const size_t lds_r = ((size_t)p.c * 2 + 256 * 2 * vec) * sizeof(float);
    const int gr = norm_bwd_reduce_grid(p.npix, p.c);
    if (dtype) {
        if (drop) hipLaunchKernelGGL((norm_bwd_reduce_kernel<BF16, true>), dim3(gr, p.nb), dim3(256), lds_r, st, p);
        else hipLaunchKernelGGL((norm_bwd_reduce_kernel<BF16, false>), dim3(gr, p.nb), dim3(256), lds_r, st, p);
    } else {
        if (drop) hipLaunchKernelGGL((norm_bwd_reduce_kernel<F32, true>), dim3(gr, p.nb), dim3(256), lds_r, st, p);
        else hipLaunchKernelGGL((norm_bwd_reduce_kernel<F32, false>), dim3(gr, p.nb), dim3(256), lds_r, st, p);
    }
    if (int r = check_launch("norm_bwd_reduce")) return r;
    int gx = norm_blocks(p.npix, p.c, vec, 8, 65535);
    if ((long long)gx * p.nb < 1024) gx = norm_blocks(p.npix, p.c, vec, 4, 65535);
    if (dtype) {
        if (drop) hipLaunchKernelGGL((norm_bwd_apply_kernel<BF16, true>), dim3(gx, p.nb), dim3(256), lds, st, p);
        else hipLaunchKernelGGL((norm_bwd_apply_kernel<BF16, false>), dim3(gx, p.nb), dim3(256), lds, st, p);
    } else {
        if (drop) hipLaunchKernelGGL((norm_bwd_apply_kernel<F32, true>), dim3(gx, p.nb), dim3(256), lds, st, p);
        else hipLaunchKernelGGL((norm_bwd_apply_kernel<F32, false>), dim3(gx, p.nb), dim3(256), lds, st, p);
    }
    return check_launch("norm_bwd_apply");
}
