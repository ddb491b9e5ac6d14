// Reductions and element-wise kernels around the network: sigmoid-Dice(+BCE) loss forward /
// backward, hard-Dice confusion counts, AdamW over a flat parameter buffer, sliding-window
// crop / blend, and small utilities.  All HBM-bound; 16-byte accesses where the layout allows.
#include <string.h>

#include <mutex>

#include "common.h"
#include "kparams.h"
#include "lossparams.h"

namespace {

__device__ __forceinline__ float load_label(const void* t, int tdtype, size_t i, int fold2) {
    float v;
    switch (tdtype) {
        case 1: v = (float)reinterpret_cast<const unsigned char*>(t)[i]; break;
        case 2: v = (float)reinterpret_cast<const long long*>(t)[i]; break;
        case 3: v = (float)reinterpret_cast<const int*>(t)[i]; break;
        default: v = reinterpret_cast<const float*>(t)[i]; break;
    }
    if (fold2 && v == 2.0f) v = 1.0f;
    return v;
}

// W consecutive elements starting at element i (a multiple of W): logits and labels.  W = 4 needs 16-byte
// aligned logits (and labels of the same width where they are loaded as one vector).
template <int W>
__device__ __forceinline__ void load_w(const float* z, size_t i, float* v) {
    if constexpr (W == 4) { const float4 q = *reinterpret_cast<const float4*>(z + i); v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w; }
    else v[0] = z[i];
}
template <int W>
__device__ __forceinline__ void load_labels(const void* t, int tdtype, size_t i, int fold2, float* v) {
    if constexpr (W == 4) {
        if (tdtype == 0) load_w<4>(reinterpret_cast<const float*>(t), i, v);
        else if (tdtype == 1) {
            const uchar4 q = *reinterpret_cast<const uchar4*>(reinterpret_cast<const unsigned char*>(t) + i);
            v[0] = (float)q.x; v[1] = (float)q.y; v[2] = (float)q.z; v[3] = (float)q.w;
        } else if (tdtype == 3) {
            const int4 q = *reinterpret_cast<const int4*>(reinterpret_cast<const int*>(t) + i);
            v[0] = (float)q.x; v[1] = (float)q.y; v[2] = (float)q.z; v[3] = (float)q.w;
        } else {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = (float)reinterpret_cast<const long long*>(t)[i + j];
        }
        if (fold2) {
#pragma unroll
            for (int j = 0; j < 4; ++j) v[j] = v[j] == 2.0f ? 1.0f : v[j];
        }
    } else {
        v[0] = load_label(t, tdtype, i, fold2);
    }
}

// sums[b] = { sum p*t, sum p, sum t, sum bce }  with p = sigmoid(z)
template <int W>
__global__ __launch_bounds__(256) void loss_reduce_kernel(const LossParams p) {
    __shared__ float red[4][4];
    const int b = blockIdx.y;
    const size_t base = (size_t)b * p.npix;
    float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
    const int ng = p.npix / W;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) {
        float zv[W], tv[W];
        load_w<W>(p.logits, base + (size_t)g * W, zv);
        load_labels<W>(p.target, p.tdtype, base + (size_t)g * W, p.fold2, tv);
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const float zi = zv[j], t = tv[j];
            const float e = __expf(-fabsf(zi));
            const float inv = __frcp_rn(1.f + e);
            const float s = zi >= 0.f ? inv : e * inv;
            a0 += s * t; a1 += s; a2 += t;
            // log(1 + e) for e in (0, 1]: the series below 2^-12 keeps the relative accuracy log1p would give
            a3 += fmaxf(zi, 0.f) - zi * t + (e < 2.44140625e-4f ? e * (1.f - 0.5f * e) : __logf(1.f + e));
        }
    }
    a0 = wave_sum(a0); a1 = wave_sum(a1); a2 = wave_sum(a2); a3 = wave_sum(a3);
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = a0; red[1][w] = a1; red[2][w] = a2; red[3][w] = a3; }
    __syncthreads();
    if (threadIdx.x < 4)
        atomicAdd(&p.sums[b * 4 + threadIdx.x],
                  (double)(((red[threadIdx.x][0] + red[threadIdx.x][1]) + red[threadIdx.x][2]) + red[threadIdx.x][3]));
}

// loss = lambda_dice * mean_b(1 - (2I+snr)/(P+T+sdr)) + lambda_ce * sum_b(bce) / (nb*npix)
// one wave: lane l takes samples l, l + 64, ..., the lanes are combined by a fixed butterfly
// loss value from the per-sample sums: lanes 0..63 of one wave
__device__ __forceinline__ void loss_finalize_body(const LossParams& p) {
    double dice = 0.0, ce = 0.0;
    for (int b = threadIdx.x; b < p.nb; b += 64) {
        const double* s = p.sums + b * 4;
        dice += 1.0 - (2.0 * s[0] + (double)p.smooth_nr) / (s[1] + s[2] + (double)p.smooth_dr);
        ce += s[3];
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) { dice += __shfl_xor(dice, o, 64); ce += __shfl_xor(ce, o, 64); }
    if (threadIdx.x != 0) return;
    const float l = (float)((double)p.lambda_dice * dice / (double)p.nb + (double)p.lambda_ce * ce / ((double)p.nb * (double)p.npix));
    p.loss[0] = l;
    if (p.loss_accum) p.loss_accum[0] += l;
}
__global__ void loss_finalize_kernel(const LossParams p) { loss_finalize_body(p); }

// dz = g * [ lambda_dice/nb * (-2 t D + (2I+snr)) / D^2 * s(1-s) + lambda_ce * (s - t)/(nb*npix) ],  D = P+T+sdr
template <int W>
__global__ __launch_bounds__(256) void loss_bwd_kernel(const LossParams p) {
    const int b = blockIdx.y;
    const size_t base = (size_t)b * p.npix;
    const float g = (p.gout ? p.gout[0] : 1.f) * p.gscale;
    const double* sm = p.sums + b * 4;
    const float D = (float)(sm[1] + sm[2] + (double)p.smooth_dr);
    const float num = (float)(2.0 * sm[0] + (double)p.smooth_nr);
    const float cd = g * p.lambda_dice / ((float)p.nb * D * D);
    const float cc = g * p.lambda_ce / ((float)p.nb * (float)p.npix);
    const int ng = p.npix / W;
    for (int g = blockIdx.x * 256 + threadIdx.x; g < ng; g += gridDim.x * 256) {
        float zv[W], tv[W], o[W];
        load_w<W>(p.logits, base + (size_t)g * W, zv);
        load_labels<W>(p.target, p.tdtype, base + (size_t)g * W, p.fold2, tv);
#pragma unroll
        for (int j = 0; j < W; ++j) {
            const float zi = zv[j], t = tv[j];
            const float e = __expf(-fabsf(zi));
            const float s = zi >= 0.f ? 1.f / (1.f + e) : e / (1.f + e);
            o[j] = cd * (num - 2.f * t * D) * s * (1.f - s) + cc * (s - t);
        }
        if constexpr (W == 4) *reinterpret_cast<float4*>(p.dlogits + base + (size_t)g * W) = make_float4(o[0], o[1], o[2], o[3]);
        else p.dlogits[base + (size_t)g] = o[0];
    }
    // one launch fewer on a training step's chain: the first wave of the grid also turns the sums into the loss value
    if (p.fuse_finalize && blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) loss_finalize_body(p);
}

// counts[b] = { tp, fp, fn, tn } as 64-bit integers; prediction = (pred >= thresh), truth = (label == 1)
__global__ __launch_bounds__(256) void dice_counts_kernel(const CountParams p) {
    __shared__ unsigned red[4][4];
    const int b = blockIdx.y;
    const float* y = p.pred + (size_t)b * p.npix;
    unsigned c0 = 0, c1 = 0, c2 = 0, c3 = 0;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < p.npix; i += gridDim.x * 256) {
        const bool pr = p.pred_is_binary ? (y[i] == 1.0f) : (y[i] >= p.thresh);
        const bool gt = load_label(p.target, p.tdtype, (size_t)b * p.npix + i, p.fold2) == 1.0f;
        c0 += pr && gt; c1 += pr && !gt; c2 += !pr && gt; c3 += !pr && !gt;
    }
    for (int o = 32; o > 0; o >>= 1) {
        c0 += __shfl_xor((int)c0, o, 64); c1 += __shfl_xor((int)c1, o, 64);
        c2 += __shfl_xor((int)c2, o, 64); c3 += __shfl_xor((int)c3, o, 64);
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = c0; red[1][w] = c1; red[2][w] = c2; red[3][w] = c3; }
    __syncthreads();
    if (threadIdx.x < 4)
        atomicAdd(&p.counts[b * 4 + threadIdx.x],
                  (unsigned long long)red[threadIdx.x][0] + red[threadIdx.x][1] + red[threadIdx.x][2] + red[threadIdx.x][3]);
}

// AdamW with decoupled weight decay over flat fp32 buffers (torch.optim.AdamW semantics,
// amsgrad off): p *= 1 - lr*wd ; m,v update ; p -= (lr/bc1) * m / (sqrt(v)/sqrt(bc2) + eps)
__global__ __launch_bounds__(256) void adamw_kernel(const AdamParams p) {
    const long long n4 = p.n / 4;
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n4; i += (long long)gridDim.x * 256) {
        float4 w = reinterpret_cast<float4*>(p.param)[i];
        float4 g = reinterpret_cast<const float4*>(p.grad)[i];
        float4 m = reinterpret_cast<float4*>(p.m)[i];
        float4 v = reinterpret_cast<float4*>(p.v)[i];
        float* wp = &w.x; float* gp = &g.x; float* mp = &m.x; float* vp = &v.x;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            const float gg = gp[e] * p.gscale;
            wp[e] *= p.decay;
            mp[e] = p.beta1 * mp[e] + (1.f - p.beta1) * gg;
            vp[e] = p.beta2 * vp[e] + (1.f - p.beta2) * gg * gg;
            wp[e] -= p.step_size * mp[e] / (sqrtf(vp[e]) * p.inv_bc2_sqrt + p.eps);
        }
        reinterpret_cast<float4*>(p.param)[i] = w;
        reinterpret_cast<float4*>(p.m)[i] = m;
        reinterpret_cast<float4*>(p.v)[i] = v;
    }
    if (blockIdx.x == 0) {
        for (long long i = n4 * 4 + threadIdx.x; i < p.n; i += 256) {
            const float gg = p.grad[i] * p.gscale;
            float w = p.param[i] * p.decay;
            const float m = p.beta1 * p.m[i] + (1.f - p.beta1) * gg;
            const float v = p.beta2 * p.v[i] + (1.f - p.beta2) * gg * gg;
            w -= p.step_size * m / (sqrtf(v) * p.inv_bc2_sqrt + p.eps);
            p.param[i] = w; p.m[i] = m; p.v[i] = v;
        }
    }
}

// windows[w][...] = padded image crop; one-channel fp32, any of 2-D / 3-D
__global__ __launch_bounds__(256) void sw_gather_kernel(const SwParams p) {
    const int w = blockIdx.y;  // window index among the launched group
    const int gw = p.win0 + w;
    const int b = gw / p.nwin, wi = gw % p.nwin;
    const int rpix = p.roi[0] * p.roi[1] * p.roi[2];
    const int* st = p.starts + wi * 3;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < rpix; i += gridDim.x * 256) {
        const int x = i % p.roi[2]; const int r = i / p.roi[2];
        const int y = r % p.roi[1]; const int z = r / p.roi[1];
        // coordinates in the padded image -> original image (zero outside: constant padding, cval 0)
        const int oz = st[0] + z - p.pad_lo[0], oy = st[1] + y - p.pad_lo[1], ox = st[2] + x - p.pad_lo[2];
        float v = p.cval;
        if ((unsigned)oz < (unsigned)p.img[0] && (unsigned)oy < (unsigned)p.img[1] && (unsigned)ox < (unsigned)p.img[2])
            v = p.image[(((size_t)b * p.img[0] + oz) * p.img[1] + oy) * p.img[2] + ox];
        p.windows[(size_t)w * rpix + i] = v;
    }
}

// out[b][pixel] = (sum over the windows covering the pixel, in window order) / count
__global__ __launch_bounds__(256) void sw_blend_kernel(const SwParams p) {
    const int b = blockIdx.y;
    const int ipix = p.img[0] * p.img[1] * p.img[2];
    const int rpix = p.roi[0] * p.roi[1] * p.roi[2];
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ipix; i += gridDim.x * 256) {
        const int x = i % p.img[2]; const int r = i / p.img[2];
        const int y = r % p.img[1]; const int z = r / p.img[1];
        const int pz = z + p.pad_lo[0], py = y + p.pad_lo[1], px = x + p.pad_lo[2];
        float acc = 0.f, cnt = 0.f;
        for (int wi = 0; wi < p.nwin; ++wi) {
            const int* st = p.starts + wi * 3;
            const int lz = pz - st[0], ly = py - st[1], lx = px - st[2];
            if ((unsigned)lz < (unsigned)p.roi[0] && (unsigned)ly < (unsigned)p.roi[1] && (unsigned)lx < (unsigned)p.roi[2]) {
                acc += p.win_out[((size_t)(b * p.nwin + wi)) * rpix + ((size_t)lz * p.roi[1] + ly) * p.roi[2] + lx];
                cnt += 1.f;
            }
        }
        p.out[(size_t)b * ipix + i] = acc / cnt;
    }
}

// dst[pix][coff..coff+c) += src[pix][...]   (channels-last, 16-byte vectors)
template <class TR>
__global__ __launch_bounds__(256) void add_kernel(char* dst, int dpitch, int dcoff, const char* src, int spitch,
                                                  int scoff, int c, long long npix) {
    constexpr int VEC = TR::VEC, ESZ = TR::ESZ;
    const int cvn = c / VEC;
    const long long nvec = npix * cvn;
    for (long long v = (long long)blockIdx.x * 256 + threadIdx.x; v < nvec; v += (long long)gridDim.x * 256) {
        const long long pix = v / cvn; const int cv = (int)(v - pix * cvn);
        float a[VEC], b[VEC];
        load_vec<TR>(dst + ((size_t)pix * dpitch + dcoff + cv * VEC) * ESZ, a);
        load_vec<TR>(src + ((size_t)pix * spitch + scoff + cv * VEC) * ESZ, b);
#pragma unroll
        for (int e = 0; e < VEC; ++e) a[e] += b[e];
        store_vec<TR>(dst + ((size_t)pix * dpitch + dcoff + cv * VEC) * ESZ, a);
    }
}

// labels == 2 -> 1, in place (reference/train.py:112-113)
__global__ __launch_bounds__(256) void fold_labels_kernel(void* t, int tdtype, long long n) {
    for (long long i = (long long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long long)gridDim.x * 256) {
        switch (tdtype) {
            case 1: { unsigned char* q = (unsigned char*)t; if (q[i] == 2) q[i] = 1; } break;
            case 2: { long long* q = (long long*)t; if (q[i] == 2) q[i] = 1; } break;
            case 3: { int* q = (int*)t; if (q[i] == 2) q[i] = 1; } break;
            default: { float* q = (float*)t; if (q[i] == 2.0f) q[i] = 1.0f; } break;
        }
    }
}

int check_launch(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string(what) + ": " + hipGetErrorString(e)); return -2; }
    return 0;
}

int grid_for(long long n, int per_block, int cap) {
    long long g = (n + per_block - 1) / per_block;
    if (g < 1) g = 1;
    if (g > cap) g = cap;
    return (int)g;
}

}  // namespace

// four pixels per lane: every sample starts on a 16-byte boundary of the logits and on a vector boundary of the labels
static bool loss_v4(const LossParams& p) {
    const int tsz = p.tdtype == 1 ? 1 : p.tdtype == 2 ? 8 : 4;
    return p.npix % 4 == 0 && ((uintptr_t)p.logits & 15) == 0 && ((uintptr_t)p.target & (uintptr_t)(4 * tsz > 16 ? 15 : 4 * tsz - 1)) == 0;
}

// sums ([nb][4] doubles) is zeroed here, then filled; loss[0] is written; loss_accum (optional) is accumulated.
int launch_loss_fwd(const LossParams& p, hipStream_t st) {
    if (!p.sums_cleared) MI_HIP(hipMemsetAsync(p.sums, 0, (size_t)p.nb * 4 * sizeof(double), st));
    if (loss_v4(p)) hipLaunchKernelGGL(loss_reduce_kernel<4>, dim3(grid_for(p.npix, 2048, 128), p.nb), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(loss_reduce_kernel<1>, dim3(grid_for(p.npix, 2048, 128), p.nb), dim3(256), 0, st, p);
    if (int r = check_launch("loss_reduce")) return r;
    if (p.fuse_finalize) return 0;        // launch_loss_bwd with the same parameters writes the loss value
    hipLaunchKernelGGL(loss_finalize_kernel, dim3(1), dim3(64), 0, st, p);
    return check_launch("loss_finalize");
}

// ---- loss forward for callers that cannot allocate ---------------------------------------------------------------
// The per-sample sums of the loss are scratch memory (partial results beyond inputs and outputs).  This variant
// keeps them in a file-scope device array instead of a caller buffer, so a caller under stream capture or inside a
// serving loop needs no workspace.  Every GPU the library runs on holds its own copy of the array at its own
// address until the process ends; host code only ever uses the address obtained per device below, never the
// array's name.  Calls on one GPU share that GPU's array: calls that overlap on one GPU race.
#define MI_NOALLOC_MAX_BATCH 4096
static __device__ __align__(256) double g_loss_scratch[MI_NOALLOC_MAX_BATCH * 4];     // 128 KiB per GPU

namespace {
constexpr int kMaxDevices = 64;
double* g_loss_scratch_ptr[kMaxDevices] = {nullptr};
std::mutex g_loss_scratch_mu;

// device address of this GPU's copy of the scratch array (looked up once per device, with that device current)
int loss_scratch_for_current_device(double** out) {
    int dev = -1;
    MI_HIP(hipGetDevice(&dev));
    MI_CHECK(dev >= 0 && dev < kMaxDevices, "device ordinal out of range");
    std::lock_guard<std::mutex> lock(g_loss_scratch_mu);
    if (!g_loss_scratch_ptr[dev]) {
        void* p = nullptr;
        MI_HIP(hipGetSymbolAddress(&p, HIP_SYMBOL(g_loss_scratch)));
        MI_CHECK(p != nullptr, "hipGetSymbolAddress returned a null address");
        g_loss_scratch_ptr[dev] = static_cast<double*>(p);
    }
    *out = g_loss_scratch_ptr[dev];
    return 0;
}
}  // namespace

extern "C" int mi_dice_ce_loss_fwd_noalloc(const float* logits, const void* target, int target_dtype, int fold2, int nb,
                                           int64_t npix, float smooth_nr, float smooth_dr, float lambda_dice, float lambda_ce,
                                           float* loss, float* loss_accum, void* stream) {
    MI_CHECK(logits && target && loss, "null argument");
    MI_CHECK(npix > 0 && npix < 2147483647LL && nb > 0, "bad sizes");
    MI_CHECK(nb <= MI_NOALLOC_MAX_BATCH, "batch exceeds the 4096 samples the built-in scratch array holds: use mi_dice_ce_loss_fwd with a caller buffer");
    double* sums = nullptr;
    if (int r = loss_scratch_for_current_device(&sums)) return r;
    LossParams p; memset(&p, 0, sizeof(p));
    p.logits = logits; p.target = target; p.tdtype = target_dtype; p.fold2 = fold2; p.sums = sums; p.loss = loss;
    p.loss_accum = loss_accum; p.nb = nb; p.npix = (int)npix; p.smooth_nr = smooth_nr; p.smooth_dr = smooth_dr;
    p.lambda_dice = lambda_dice; p.lambda_ce = lambda_ce;
    return launch_loss_fwd(p, (hipStream_t)stream);       // clears sums[0 .. 4*nb) on the stream, then accumulates
}

int launch_loss_bwd(const LossParams& p, hipStream_t st) {
    if (loss_v4(p) && ((uintptr_t)p.dlogits & 15) == 0)
        hipLaunchKernelGGL(loss_bwd_kernel<4>, dim3(grid_for(p.npix, 2048, 256), p.nb), dim3(256), 0, st, p);
    else hipLaunchKernelGGL(loss_bwd_kernel<1>, dim3(grid_for(p.npix, 1024, 256), p.nb), dim3(256), 0, st, p);
    return check_launch("loss_bwd");
}

int launch_dice_counts(const CountParams& p, hipStream_t st) {
    MI_HIP(hipMemsetAsync(p.counts, 0, (size_t)p.nb * 4 * sizeof(unsigned long long), st));
    hipLaunchKernelGGL(dice_counts_kernel, dim3(grid_for(p.npix, 2048, 128), p.nb), dim3(256), 0, st, p);
    return check_launch("dice_counts");
}

int launch_adamw(const AdamParams& p, hipStream_t st) {
    hipLaunchKernelGGL(adamw_kernel, dim3(grid_for(p.n / 4, 256, 2048)), dim3(256), 0, st, p);
    return check_launch("adamw");
}

int launch_sw_gather(const SwParams& p, int nwin_launch, hipStream_t st) {
    const int rpix = p.roi[0] * p.roi[1] * p.roi[2];
    hipLaunchKernelGGL(sw_gather_kernel, dim3(grid_for(rpix, 1024, 256), nwin_launch), dim3(256), 0, st, p);
    return check_launch("sw_gather");
}

int launch_sw_blend(const SwParams& p, hipStream_t st) {
    const int ipix = p.img[0] * p.img[1] * p.img[2];
    hipLaunchKernelGGL(sw_blend_kernel, dim3(grid_for(ipix, 1024, 256), p.nb), dim3(256), 0, st, p);
    return check_launch("sw_blend");
}

int launch_add(int dtype, char* dst, int dpitch, int dcoff, const char* src, int spitch, int scoff, int c,
               long long npix, hipStream_t st) {
    const int vec = dtype ? 8 : 4;
    MI_CHECK(c % vec == 0, "channels must be a multiple of the 16-byte vector");
    const int g = grid_for(npix * (c / vec), 1024, 2048);
    if (dtype) hipLaunchKernelGGL(add_kernel<BF16>, dim3(g), dim3(256), 0, st, dst, dpitch, dcoff, src, spitch, scoff, c, npix);
    else hipLaunchKernelGGL(add_kernel<F32>, dim3(g), dim3(256), 0, st, dst, dpitch, dcoff, src, spitch, scoff, c, npix);
    return check_launch("add");
}

int launch_fold_labels(void* t, int tdtype, long long n, hipStream_t st) {
    hipLaunchKernelGGL(fold_labels_kernel, dim3(grid_for(n, 1024, 1024)), dim3(256), 0, st, t, tdtype, n);
    return check_launch("fold_labels");
}
