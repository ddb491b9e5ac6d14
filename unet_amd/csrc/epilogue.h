// Pieces shared by the gather-GEMM kernels of igemm.hip and igemm_dma.hip: the MFMA wrapper, the channel order of
// the accumulator rows, the 16-lane row sum and the tile epilogue (statistics, bias, addend, two destinations, raw
// partial sums under a contraction split).
#pragma once
#include "common.h"
#include "kparams.h"

namespace {

template <class TR>
__device__ __forceinline__ v4f mma16(const uint4& a, const uint4& b, v4f c);
template <>
__device__ __forceinline__ v4f mma16<BF16>(const uint4& a, const uint4& b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ v4f mma16<F32>(const uint4& a, const uint4& b, v4f c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
}

// n / d for 0 <= n < 2^32 / d through the host-computed reciprocal mg = 2^32 / d + 1 (mg == 0: d == 1)
__device__ __forceinline__ int fast_div(int n, unsigned mg) { return mg ? (int)__umulhi((unsigned)n, mg) : n; }

// channel (within the workgroup's N tile) held by MFMA row `rr` of fragment `nf`
template <int NFRAG>
__device__ __forceinline__ int chan_of_row(int nf, int rr) {
    if (NFRAG == 1) return rr;
    return 32 * (nf >> 1) + 8 * (rr >> 2) + 4 * (nf & 1) + (rr & 3);
}

// sum over the 16 lanes of a DPP row (all lanes receive it): two quad permutes, half mirror, mirror
__device__ __forceinline__ float row_sum16(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));   // quad_perm [1,0,3,2]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));   // quad_perm [2,3,0,1]
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));  // row_half_mirror
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));  // row_mirror
    return v;
}


// Epilogue of one tile of 256 base pixels (pixel pidx = 64 * wave + 16 * mf + lr of the tile, four consecutive output
// channels per accumulator register): instance-norm statistics of the bias-free result, bias, addend, store to one of
// two destinations; under a contraction split the raw fp32 partial sums go to p.kpart instead.  smem is reused for the
// statistics partials: nothing else may still be reading or writing it when the workgroup gets here.
template <class TR, int NFRAG>
__device__ __forceinline__ void igemm_epilogue(const IgemmParams& p, v4f (&acc)[4][NFRAG], char* smem, int img, int slot, int n0, int ph,
                                               int ksp, int bz0, int by0, int bx0) {
    constexpr int ESZ = TR::ESZ;
    constexpr int NT = NFRAG * 16;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    const int mx = (1 << p.tlog[2]) - 1, my = (1 << p.tlog[1]) - 1;
    const bool do_stats = p.stats != nullptr && p.ksplit <= 1;
    float s1[NFRAG][4], s2[NFRAG][4];
#pragma unroll
    for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[nf][j] = 0.f; s2[nf][j] = 0.f; }

#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        const int pidx = wave * 64 + mf * 16 + lr;
        const int tx = pidx & mx, ty = (pidx >> p.tlog[2]) & my, tz = pidx >> (p.tlog[2] + p.tlog[1]);
        const int bz = bz0 + tz, by = by0 + ty, bx = bx0 + tx;
        const bool valid = bz < p.bd[0] && by < p.bd[1] && bx < p.bd[2];
        if (do_stats) {
#pragma unroll
            for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = valid ? acc[mf][nf][j] : 0.f;
                    s1[nf][j] += v; s2[nf][j] += v * v;
                }
        }
        if (valid && p.ksplit > 1) {
            // contraction split: raw fp32 partial sums [split][output pixel][N]; the finishing kernel adds them up
            const int oz = bz * p.so + p.phase_off[ph][0], oy = by * p.so + p.phase_off[ph][1],
                      ox = bx * p.so + p.phase_off[ph][2];
            const size_t op = ((size_t)(img * p.od[0] + oz) * p.od[1] + oy) * p.od[2] + ox;
            const size_t npo = (size_t)p.nb * p.od[0] * p.od[1] * p.od[2];
            if (p.kpart_blocked) {
                // [split][image][N / 8][pixel of the image][8]: the reader owns one image and eight channels
                const size_t npi = (size_t)p.od[0] * p.od[1] * p.od[2];
                const size_t pin = ((size_t)oz * p.od[1] + oy) * p.od[2] + ox;
                float* dst = p.kpart + (((size_t)ksp * p.nb + img) * (p.N / 8)) * npi * 8 + pin * 8;
#pragma unroll
                for (int nf = 0; nf < NFRAG; ++nf) {
                    const int c = n0 + chan_of_row<NFRAG>(nf, 4 * g);
                    *reinterpret_cast<float4*>(dst + (size_t)(c >> 3) * npi * 8 + (c & 7)) =
                        make_float4(acc[mf][nf][0], acc[mf][nf][1], acc[mf][nf][2], acc[mf][nf][3]);
                }
            } else {
                float* dst = p.kpart + ((size_t)ksp * npo + op) * p.N + n0;
#pragma unroll
                for (int nf = 0; nf < NFRAG; ++nf) {
                    const int tc = chan_of_row<NFRAG>(nf, 4 * g);     // four consecutive channels
                    *reinterpret_cast<float4*>(dst + tc) = make_float4(acc[mf][nf][0], acc[mf][nf][1], acc[mf][nf][2], acc[mf][nf][3]);
                }
            }
        } else if (valid) {
            const int oz = bz * p.so + p.phase_off[ph][0], oy = by * p.so + p.phase_off[ph][1],
                      ox = bx * p.so + p.phase_off[ph][2];
            const size_t op = ((size_t)(img * p.od[0] + oz) * p.od[1] + oy) * p.od[2] + ox;
            if (NFRAG == 1) {
                const int ch = n0 + 4 * g;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mf][0][j];
                if (p.bias && ch >= p.bias_c0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += p.bias[ch - p.bias_c0 + j];
                }
                if (p.addend) {
                    float a[4];
                    load_n<TR, 4>(p.addend + (op * p.add_pitch + p.add_coff + ch) * ESZ, a);
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += a[j];
                }
                char* dst = ch < p.out0_c ? p.out + (op * p.out_pitch + p.out_coff + ch) * ESZ
                                          : p.out1 + (op * p.out1_pitch + p.out1_coff + ch - p.out0_c) * ESZ;
                store_n<TR, 4>(dst, v);
            } else {
#pragma unroll
                for (int pf = 0; pf < NFRAG / 2; ++pf) {
                    const int ch = n0 + 32 * pf + 8 * g;
                    float v[8];
#pragma unroll
                    for (int j = 0; j < 4; ++j) { v[j] = acc[mf][2 * pf][j]; v[4 + j] = acc[mf][2 * pf + 1][j]; }
                    if (p.bias && ch >= p.bias_c0) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += p.bias[ch - p.bias_c0 + j];
                    }
                    if (p.addend) {
                        float a[8];
                        load_n<TR, 8>(p.addend + (op * p.add_pitch + p.add_coff + ch) * ESZ, a);
#pragma unroll
                        for (int j = 0; j < 8; ++j) v[j] += a[j];
                    }
                    char* dst = ch < p.out0_c ? p.out + (op * p.out_pitch + p.out_coff + ch) * ESZ
                                              : p.out1 + (op * p.out1_pitch + p.out1_coff + ch - p.out0_c) * ESZ;
                    store_n<TR, 8>(dst, v);
                }
            }
        }
    }
    if (do_stats) {
        // fixed-order reduction inside the workgroup (16-lane DPP row sums, then the four waves in
        // order); the per-workgroup sums are added to fp64 accumulators, so the result does not
        // depend on the order in which workgroups arrive (to fp32 rounding)
        __syncthreads();
        float* sst = reinterpret_cast<float*>(smem);  // [4 waves][NT][2]
#pragma unroll
        for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = row_sum16(s1[nf][j]), b = row_sum16(s2[nf][j]);
                if (lr == 0) {
                    const int cl = chan_of_row<NFRAG>(nf, 4 * g + j);
                    sst[(wave * NT + cl) * 2] = a;
                    sst[(wave * NT + cl) * 2 + 1] = b;
                }
            }
        __syncthreads();
        for (int i = tid; i < NT * 2; i += 256) {
            const int ch = n0 + (i >> 1);
            if (ch < p.nstat) {
                const float v = ((sst[i] + sst[NT * 2 + i]) + sst[2 * NT * 2 + i]) + sst[3 * NT * 2 + i];
                atomicAdd(&p.stats[(((size_t)img * MI_NSLOT + slot) * p.stats_c + ch) * 2 + (i & 1)], (double)v);
            }
        }
    }
}

}  // namespace
