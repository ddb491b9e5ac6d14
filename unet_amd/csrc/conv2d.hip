// Two-dimensional gather-GEMM convolutions with asynchronous staging (gfx950).
//
// Same contract as igemm_kernel (kparams.h, IgemmParams) for 2-D tensors and a 16 x 16 base-pixel tile.  Five kernels:
//   conv2d_dma_kernel     unit stride, 3 x 3 taps (18 x 18 patch) or one tap (16 x 16)
//   conv2d_phase_kernel   the four stride-2 scatter phases (transposed convolution, data gradient of a stride-2
//                         convolution) from one 17 x 17 patch
//   conv2d_s2_kernel      stride-2 gather (stride-2 convolution, data gradient of the transposed convolution) as four
//                         polyphase component stages
//   conv2d_k16_kernel, conv2d_s2k16_kernel   the unit-stride and stride-2 gather forms for 16 bf16 contraction
//                         channels: 32-byte pixel rows, two taps per MFMA contraction step
// What they share, and what differs from the register-staged kernels of igemm.hip:
//
//  * the input patch and the weight tile of a stage go from global memory straight into LDS (global_load ... lds,
//    16 bytes per lane, no registers in between), the loads of stage s + 1 are issued before the MFMA loop of
//    stage s and land while it runs: one barrier per stage, and a workgroup that walks several tiles has the next
//    tile's patch in flight during the current tile's arithmetic and epilogue;
//  * LDS rows are unpadded 64-byte pixel (or weight) rows; the 16-byte chunk c of row r sits in slot
//    c ^ (((x >> 2) & 1) << 1), x = the pixel's column in the patch (the row number for weights).  The DMA writes LDS
//    lane-linearly, so the permutation is applied to the per-lane SOURCE address, and again on the fragment reads;
//    with it the 16 lanes of every ds_read_b128 service group hit 16 different 16-byte slots for any tap offset
//    (checked exhaustively on the host, tests/test_host.py); 32-byte rows need no permutation;
//  * pixels outside the tensor are read from a zero page, so the staged image needs no separate clearing pass.
//
// Stages of conv2d_dma_kernel: with resident weights (all contraction chunks of all taps fit next to two patch
// buffers) a stage is one (tile, chunk) and multiplies all taps; otherwise a stage is one kernel row of a
// (tile, chunk) whose weights go through a ring of three buffers.
#include "common.h"
#include "kparams.h"

namespace {

static __device__ __align__(256) unsigned char g_zero_page[256];

template <class TR>
__device__ __forceinline__ v4f mma16v(const uint4& a, const uint4& b, v4f c);
template <>
__device__ __forceinline__ v4f mma16v<BF16>(const uint4& a, const uint4& b, v4f c) {
    return __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, a), __builtin_bit_cast(v8bf, b), c, 0, 0, 0);
}
template <>
__device__ __forceinline__ v4f mma16v<F32>(const uint4& a, const uint4& b, v4f c) {
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.x), __uint_as_float(b.x), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.y), __uint_as_float(b.y), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.z), __uint_as_float(b.z), c, 0, 0, 0);
    c = __builtin_amdgcn_mfma_f32_16x16x4f32(__uint_as_float(a.w), __uint_as_float(b.w), c, 0, 0, 0);
    return c;
}

__device__ __forceinline__ float row_sum16v(float v) {
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0xB1, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x4E, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x141, 0xf, 0xf, true));
    v += __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), 0x140, 0xf, 0xf, true));
    return v;
}

template <int NFRAG>
__device__ __forceinline__ int chan_of_row_v(int nf, int rr) {
    if (NFRAG == 1) return rr;
    return 32 * (nf >> 1) + 8 * (rr >> 2) + 4 * (nf & 1) + (rr & 3);
}

// 16 bytes per lane from global memory to LDS at lds_base + 16 * lane (lds_base wave-uniform)
__device__ __forceinline__ void dma16(const char* gsrc, char* lds_base) {
    // Inline assembly on purpose: hipcc drains every outstanding LDS-DMA it knows of (s_waitcnt vmcnt(0)) in front
    // of the next LDS read, which would undo the prefetch; loads it does not know of are waited for by the counted
    // s_waitcnt vmcnt(N) of the stage loops alone.  M0 holds the wave-uniform LDS destination and is restored.
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

__device__ __forceinline__ int swz(int row) { return ((row >> 2) & 1) << 1; }


// Epilogue of one 16 x 16 base-pixel tile: the wave's four accumulator rows (base rows 4 * wave + mf, column lr) get
// bias / addend, go to one of the two destinations, and feed the running norm statistics.  Output pixel of base
// pixel (y, x): base_pix + rel0 + mf * rel_step (the caller folds in image, tile origin, output stride and phase).
// ADDM: 0 = the addend is a run-time option, loaded row by row; 1 = the caller knows there is none; 2 = there is one
// and all four rows' addend vectors are requested before the first store (the addend may be the destination itself, a
// gradient accumulated in place, so left alone the compiler keeps every addend load behind the previous row's store).
// G3: the tile is a 3-D one (wave = z plane, mf = y row): by0 then carries the validity of the wave's four rows as a
// bit mask and bx0 that of the lane's column.
template <class TR, int NFRAG, int ADDM = 0, bool G3 = false>
__device__ __forceinline__ void tile_epilogue(const IgemmParams& p, const v4f (&acc)[4][NFRAG], int base_pix, int rel0, int rel_step,
                                              int by0, int bx0, int wave, int lr, int g, int n0, bool do_stats,
                                              float (&s1)[NFRAG][4], float (&s2)[NFRAG][4]) {
    constexpr int ESZ = TR::ESZ;
    const bool full = !G3 && by0 + 16 <= p.bd[1] && bx0 + 16 <= p.bd[2];
    constexpr int NV = NFRAG == 1 ? 1 : NFRAG / 2, CW = NFRAG == 1 ? 4 : 8, AW = CW * ESZ / 4;
    unsigned av[4][NV][AW];                             // ADDM == 2: packed addend vectors of the four rows
    if (ADDM == 2) {
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const bool valid = G3 ? (((by0 >> mf) & 1) && bx0) : (full || (by0 + wave * 4 + mf < p.bd[1] && bx0 + lr < p.bd[2]));
            const size_t op = (size_t)(base_pix + (valid ? rel0 + mf * rel_step : 0));     // outside the tensor: the tile's origin, unused
#pragma unroll
            for (int pf = 0; pf < NV; ++pf) {
                const int ch = NFRAG == 1 ? n0 + 4 * g : n0 + 32 * pf + 8 * g;
                const char* src = p.addend + (op * p.add_pitch + p.add_coff + ch) * ESZ;
                if constexpr (AW == 2) {
                    const uint2 t = *reinterpret_cast<const uint2*>(src);
                    av[mf][pf][0] = t.x; av[mf][pf][1] = t.y;
                } else {
#pragma unroll
                    for (int q = 0; q < AW / 4; ++q) {
                        const uint4 t = *reinterpret_cast<const uint4*>(src + 16 * q);
                        av[mf][pf][4 * q] = t.x; av[mf][pf][4 * q + 1] = t.y; av[mf][pf][4 * q + 2] = t.z; av[mf][pf][4 * q + 3] = t.w;
                    }
                }
            }
        }
    }
    auto add_kept = [&](float* v, int mf, int pf) {
#pragma unroll
        for (int j = 0; j < CW; ++j) {
            if constexpr (ESZ == 2) v[j] += __uint_as_float((j & 1) ? (av[mf][pf][j >> 1] & 0xffff0000u) : (av[mf][pf][j >> 1] << 16));
            else v[j] += __uint_as_float(av[mf][pf][j]);
        }
    };
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        const bool valid = G3 ? (((by0 >> mf) & 1) && bx0) : (full || (by0 + wave * 4 + mf < p.bd[1] && bx0 + lr < p.bd[2]));
        if (do_stats) {
#pragma unroll
            for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const float v = valid ? acc[mf][nf][j] : 0.f;
                    s1[nf][j] += v; s2[nf][j] += v * v;
                }
        }
        if (valid) {
            const size_t op = (size_t)(base_pix + rel0 + mf * rel_step);
            if (NFRAG == 1) {
                const int ch = n0 + 4 * g;
                float v[4];
#pragma unroll
                for (int j = 0; j < 4; ++j) v[j] = acc[mf][0][j];
                if (p.bias && ch >= p.bias_c0) {
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] += p.bias[ch - p.bias_c0 + j];
                }
                if (ADDM == 2) add_kept(v, mf, 0);
                if (ADDM == 0 && p.addend) {
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
                    if (ADDM == 2) add_kept(v, mf, pf);
                    if (ADDM == 0 && p.addend) {
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
}

// Adds the workgroup's statistics partials to the fp64 accumulators of image img_wg (all tiles of a workgroup lie in
// one image).  smem is reused as [4 waves][NT][2] floats: every LDS-DMA must have landed and been consumed.
template <int NFRAG>
__device__ __forceinline__ void flush_stats(const IgemmParams& p, char* smem, const float (&s1)[NFRAG][4], const float (&s2)[NFRAG][4],
                                            int wave, int lr, int g, int n0, int img_wg, int slot) {
    constexpr int NT = NFRAG * 16;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    float* sst = reinterpret_cast<float*>(smem);
#pragma unroll
    for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const float a = row_sum16v(s1[nf][j]), bsum = row_sum16v(s2[nf][j]);
            if (lr == 0) {
                const int cl = chan_of_row_v<NFRAG>(nf, 4 * g + j);
                sst[(wave * NT + cl) * 2] = a;
                sst[(wave * NT + cl) * 2 + 1] = bsum;
            }
        }
    __syncthreads();
    for (int i = threadIdx.x; i < NT * 2; i += 256) {
        const int ch = n0 + (i >> 1);
        if (ch < p.nstat) {
            const float v = ((sst[i] + sst[NT * 2 + i]) + sst[2 * NT * 2 + i]) + sst[3 * NT * 2 + i];
            atomicAdd(&p.stats[(((size_t)img_wg * MI_NSLOT + slot) * p.stats_c + ch) * 2 + (i & 1)], (double)v);
        }
    }
}

// KS: 3 (3 x 3 taps, 18 x 18 patch) or 1 (one tap, 16 x 16).  TG: taps per stage; WRES: weights of all chunks and
// taps stay in LDS (TG must then be the tap count).
// all but the N youngest vector-memory operations of this wave have completed
template <int N>
__device__ __forceinline__ void wait_vmcnt_c() {
    static_assert(N >= 0 && N < 64, "vmcnt is a 6-bit field");
    asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N) : "memory");
}

// KS: 3 (3 x 3 taps in patch order, 18 x 18 patch) or 1 (one tap, 16 x 16).
// WRES: the weights of all chunks and taps stay in LDS and a stage is one (tile, chunk) with all taps; otherwise a
// stage is one kernel row of one (tile, chunk), its weights in a ring of three buffers, and the loads of the stage
// after next are issued at the top of a stage (one stage stays in flight across every barrier).
// The loop body holds no integer division, no address arithmetic per fragment read (three per-lane bases per patch
// buffer, everything else is an immediate offset) and every patch row fragment is read once per chunk and used by
// up to three kernel rows: the MFMA share of the issued instructions decides the speed of this kernel.
template <class TR, int NFRAG, int KS, bool WRES>
__global__ __launch_bounds__(256, 2) void conv2d_dma_kernel(const IgemmParams p) {
    signal_start(p.start_flag, p.start_val);
    constexpr int ESZ = TR::ESZ;
    constexpr int KCE = 64 / ESZ;                  // contraction elements per 64-byte chunk row
    constexpr int NT = NFRAG * 16;
    constexpr int PD = KS == 3 ? 18 : 16;
    constexpr int PP = PD * PD;
    constexpr int NIP = (PP + 15) / 16;            // DMA wave-instructions per patch (16 pixel rows each)
    constexpr int PBYTES = NIP * 1024;
    constexpr int NUP = (NIP + 3) / 4;             // ... per wave
    constexpr int NTAP = KS * KS;
    constexpr int NG = WRES ? 1 : KS;              // stages per chunk
    constexpr int TG = NTAP / NG;                  // taps per stage
    constexpr int NWB = 3;                         // weight ring (streamed weights)
    constexpr int WROWS = TG * NT;                 // weight rows per stage
    constexpr int NIW = WROWS / 16;                // DMA wave-instructions per weight stage
    constexpr int NUW = (NIW + 3) / 4;
    constexpr int WBYTES = WROWS * 64;
    constexpr int NROW = 4 + KS - 1;               // patch rows a wave touches
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const smP = smem;                        // two patch buffers
    char* const smW = smem + 2 * PBYTES;           // WRES: [chunk][tap][n] rows; else a ring of NWB stage buffers

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    const int H = p.id[1], W = p.id[2];
    const int ntx = p.ntile[2], nty = p.ntile[1];
    const int tile0 = blockIdx.x * p.tpw;
    const int ntile_wg = p.tpw;                    // the launcher makes tpw divide the tiles of an image
    const int n0 = blockIdx.y * NT;
    const int slot = blockIdx.x % MI_NSLOT;
    const int nkc = p.K / KCE;
    const int nchunk = ntile_wg * nkc;
    const int spb0 = p.in0_pitch * ESZ, spb1 = p.in1_pitch * ESZ;
    const int nkc0 = min(nkc, p.in0_c / KCE);      // chunks that come from the first source

    // ---- per-lane DMA constants ------------------------------------------------------------------------------------
    // patch instruction i covers the 16 patch pixels 16 i .. 16 i + 15 (four lanes per pixel, one 16-byte slot each);
    // slot j of pixel (py, px) receives source chunk j ^ swz(px)
    int d_off0[NUP], d_off1[NUP], d_yx[NUP];
#pragma unroll
    for (int u = 0; u < NUP; ++u) {
        const int pix = ((wave + 4 * u) % NIP) * 16 + (lane >> 2);
        const int py = pix / PD, px = pix - py * PD;
        const int cb = ((lane & 3) ^ swz(px)) * 16;
        d_off0[u] = (py * W + px) * spb0 + cb;
        d_off1[u] = (py * W + px) * spb1 + cb;
        d_yx[u] = pix < PP ? ((py << 16) | px) : -1;
    }
    int w_off[NUW];
#pragma unroll
    for (int u = 0; u < NUW; ++u) {
        const int row = ((wave + 4 * u) % NIW) * 16 + (lane >> 2);
        const int tl = row / NT, r = row - tl * NT;
        const int n = n0 + chan_of_row_v<NFRAG>(r >> 4, r & 15);
        w_off[u] = ((tl * p.N + n) * p.K) * ESZ + ((lane & 3) ^ swz(row)) * 16;
    }
    const char* const zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 3) * 16;

    // position of the next chunk whose loads are issued: (image, tile row, tile column, chunk) and running counters
    int q_img, q_ty, q_tx, q_kc = 0, q_chunk = 0, q_tg = 0;
    {
        const int tiles_img = ntx * nty;
        q_img = tile0 / tiles_img;
        const int r = tile0 - q_img * tiles_img;
        q_ty = r / ntx; q_tx = r - q_ty * ntx;
    }
    int c_img = q_img, c_ty = q_ty, c_tx = q_tx;   // the same for the chunk being multiplied
    const int img_wg = q_img;

    auto issue_patch = [&]() {
        const int iy0 = (q_ty << 4) + p.omin[1], ix0 = (q_tx << 4) + p.omin[2];
        const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + PD <= H && ix0 + PD <= W;
        const bool first = q_kc < nkc0;
        const char* pb = first ? p.in0 + (((long long)q_img * H + iy0) * W + ix0) * spb0 + (p.in0_coff + q_kc * KCE) * ESZ
                               : p.in1 + (((long long)q_img * H + iy0) * W + ix0) * spb1 + (p.in1_coff + (q_kc - nkc0) * KCE) * ESZ;
        char* dst = smP + (q_chunk & 1) * PBYTES;
#pragma unroll
        for (int u = 0; u < NUP; ++u) {
            const int i = (wave + 4 * u) % NIP;      // past the end: a harmless repeat, so every wave issues NUP loads
            const int off = first ? d_off0[u] : d_off1[u];
            if (interior) {
                dma16(d_yx[u] >= 0 ? pb + off : zero, dst + i * 1024);
            } else {
                bool ok = d_yx[u] >= 0;
                const int iy = iy0 + (d_yx[u] >> 16), ix = ix0 + (d_yx[u] & 0xffff);
                ok = ok && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                dma16(ok ? pb + off : zero, dst + i * 1024);
            }
        }
    };
    auto issue_weights = [&](int kc, int tg, char* dst) {
        const char* wb = p.w + ((size_t)(tg * TG) * p.N * p.K + (size_t)kc * KCE) * ESZ;
#pragma unroll
        for (int u = 0; u < NUW; ++u) {
            const int i = (wave + 4 * u) % NIW;      // repeats past the end, as for the patch
            dma16(wb + w_off[u], dst + i * 1024);
        }
    };
    // issues the loads of the next stage in line (nothing once every chunk has been issued) and advances the position
    int q_ring = 0;
    auto issue_next = [&]() {
        if (q_chunk < nchunk) {
            if (q_tg == 0) issue_patch();
            if (!WRES) issue_weights(q_kc, q_tg, smW + q_ring * WBYTES);
        }
        if (!WRES) q_ring = q_ring == NWB - 1 ? 0 : q_ring + 1;
        if (++q_tg == NG) {
            q_tg = 0; ++q_chunk;
            if (++q_kc == nkc) { q_kc = 0; if (++q_tx == ntx) { q_tx = 0; if (++q_ty == nty) { q_ty = 0; ++q_img; } } }
        }
    };

    // ---- per-lane fragment read bases --------------------------------------------------------------------------------
    int bx[KS];                                    // patch row 4 * wave, column lr + dx, chunk g (swizzled), buffer 0
#pragma unroll
    for (int dx = 0; dx < KS; ++dx) bx[dx] = ((wave * 4) * PD + lr + dx) * 64 + ((g ^ swz(lr + dx)) << 4);
    const int w_lane = lr * 64 + ((g ^ swz(lr)) << 4);               // NT is a multiple of 16: the row's swizzle is lr's
    const int rel0 = (wave * 4 * p.so) * p.od[2] + lr * p.so;
    const int rel_step = p.so * p.od[2];

    const bool do_stats = p.stats != nullptr;
    float s1[NFRAG][4], s2[NFRAG][4];
#pragma unroll
    for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[nf][j] = 0.f; s2[nf][j] = 0.f; }

    // ---- prologue ------------------------------------------------------------------------------------------------------
    constexpr int LA = WRES ? 1 : 2;               // stages issued ahead
    if (WRES)
        for (int c = 0; c < nkc; ++c) issue_weights(c, 0, smW + c * WBYTES);
#pragma unroll
    for (int j = 0; j < LA; ++j) issue_next();

    v4f acc[4][NFRAG];
    int kc = 0, c_ring = 0;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        int cur[KS];
        const int poff = (chunk & 1) * PBYTES;
#pragma unroll
        for (int dx = 0; dx < KS; ++dx) cur[dx] = bx[dx] + poff;
        if (kc == 0) {
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = v4f{0.f, 0.f, 0.f, 0.f};
        }
        uint4 B[NROW][KS];                         // patch row fragments of this chunk (each read once)
#pragma unroll
        for (int tgc = 0; tgc < NG; ++tgc) {
            // this wave's loads of the stage after this one may stay in flight; everything older must have landed
            if (LA == 1) {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else if (tgc == NG - 1) {            // the next stage opens a chunk: weights and a patch
                if (chunk + 1 < nchunk) wait_vmcnt_c<NUW + NUP>(); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            } else {
                wait_vmcnt_c<NUW>();
            }
            __builtin_amdgcn_s_barrier();          // this stage has landed everywhere; the previous one has been consumed
            asm volatile("" ::: "memory");
            const char* wbuf = WRES ? smW + kc * WBYTES : smW + ((NG == NWB) ? tgc : c_ring) * WBYTES;
            // rows this stage adds to the register window: all of them (whole chunk), or rows 0..3 / 4 / 5
#pragma unroll
            for (int r = 0; r < NROW; ++r) {
                const bool load = WRES ? true : (tgc == 0 ? r < 4 : r == 3 + tgc);
                if (load) {
#pragma unroll
                    for (int dx = 0; dx < KS; ++dx)
                        B[r][dx] = *reinterpret_cast<const uint4*>(smem + cur[dx] + r * PD * 64);
                }
            }
            uint4 afr[TG][NFRAG];
#pragma unroll
            for (int nf = 0; nf < NFRAG; ++nf) afr[0][nf] = *reinterpret_cast<const uint4*>(wbuf + (nf * 16) * 64 + w_lane);
            // the fragment reads above are in flight while the loads of the stage LA ahead are issued
            issue_next();
#pragma unroll
            for (int tl = 0; tl < TG; ++tl) {
                const int t = tgc * TG + tl;       // tap in patch order: kernel row t / KS, column t % KS
                const int dy = KS == 3 ? t / 3 : 0, dx = KS == 3 ? t % 3 : 0;
                if (tl + 1 < TG) {
#pragma unroll
                    for (int nf = 0; nf < NFRAG; ++nf)
                        afr[tl + 1][nf] = *reinterpret_cast<const uint4*>(wbuf + ((tl + 1) * NT + nf * 16) * 64 + w_lane);
                }
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                    for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = mma16v<TR>(afr[tl][nf], B[mf + dy][dx], acc[mf][nf]);
            }
            if (!WRES && NG != NWB) c_ring = c_ring == NWB - 1 ? 0 : c_ring + 1;
        }

        if (++kc == nkc) {
            kc = 0;
            // ---- tile epilogue ---------------------------------------------------------------------------------
            const int by0 = c_ty << 4, bx0 = c_tx << 4;
            const int base_pix = (c_img * p.od[1] + by0 * p.so + p.phase_off[0][1]) * p.od[2] + bx0 * p.so + p.phase_off[0][2];
            if (++c_tx == ntx) { c_tx = 0; if (++c_ty == nty) { c_ty = 0; ++c_img; } }
            tile_epilogue<TR, NFRAG>(p, acc, base_pix, rel0, rel_step, by0, bx0, wave, lr, g, n0, do_stats, s1, s2);
        }
    }

    if (do_stats) flush_stats<NFRAG>(p, smem, s1, s2, wave, lr, g, n0, img_wg, slot);
}

// ---- stride-2 scatter ("phase") form: transposed convolution k3 s2, and the data gradient of a stride-2 convolution --
// Base pixel (y, x) of the gathered tensor produces the four output pixels (2y + a, 2x + b); phase (a, b) sums
// 1, 2, 2 or 4 taps at gather offsets in {0, 1}^2.  One stage = one (tile, chunk): a 17 x 17 patch (unit gather stride)
// and the nine tap weight tiles of the chunk, both double-buffered, loads of the next chunk issued under the MFMA loop.
// A patch fragment is read once per chunk (5 rows x 2 columns per wave) and feeds all phases that touch it.
// tap slots in the order the host emits them: phase (0,0): (0,0); (0,1): (0,1) (0,0); (1,0): (1,0) (0,0);
// (1,1): (1,1) (1,0) (0,1) (0,0)    [(row offset, column offset)]
__host__ __device__ constexpr int ph_of_tap(int t) { return t < 1 ? 0 : t < 3 ? 1 : t < 5 ? 2 : 3; }
__host__ __device__ constexpr int ph_dy(int t) { return (t == 3 || t == 5 || t == 6) ? 1 : 0; }
__host__ __device__ constexpr int ph_dx(int t) { return (t == 1 || t == 5 || t == 7) ? 1 : 0; }

template <class TR, int NFRAG, bool ADD>     // ADD: the layer has an addend (a kernel of its own: the other keeps the shorter epilogue)
__global__ __launch_bounds__(256, 2) void conv2d_phase_kernel(const IgemmParams p) {
    signal_start(p.start_flag, p.start_val);
    constexpr int ESZ = TR::ESZ;
    constexpr int KCE = 64 / ESZ;
    constexpr int NT = NFRAG * 16;
    constexpr int PD = 17, PP = PD * PD;
    constexpr int NIP = (PP + 15) / 16;            // 19 DMA wave-instructions per patch
    constexpr int PBYTES = NIP * 1024;
    constexpr int NUP = (NIP + 3) / 4;
    constexpr int NTAP = 9;
    constexpr int WROWS = NTAP * NT;
    constexpr int NIW = WROWS / 16;
    constexpr int NUW = (NIW + 3) / 4;
    constexpr int WBYTES = WROWS * 64;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const smP = smem;                        // two patch buffers
    char* const smW = smem + 2 * PBYTES;           // two weight buffers [tap][n] rows

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    const int H = p.id[1], W = p.id[2];
    const int ntx = p.ntile[2], nty = p.ntile[1];
    const int tile0 = blockIdx.x * p.tpw;
    const int n0 = blockIdx.y * NT;
    const int slot = blockIdx.x % MI_NSLOT;
    const int nkc = p.K / KCE;
    const int nchunk = p.tpw * nkc;
    const int spb0 = p.in0_pitch * ESZ, spb1 = p.in1_pitch * ESZ;
    const int nkc0 = min(nkc, p.in0_c / KCE);

    int d_off0[NUP], d_off1[NUP], d_yx[NUP];
#pragma unroll
    for (int u = 0; u < NUP; ++u) {
        const int pix = ((wave + 4 * u) % NIP) * 16 + (lane >> 2);
        const int py = pix / PD, px = pix - py * PD;
        const int cb = ((lane & 3) ^ swz(px)) * 16;
        d_off0[u] = (py * W + px) * spb0 + cb;
        d_off1[u] = (py * W + px) * spb1 + cb;
        d_yx[u] = pix < PP ? ((py << 16) | px) : -1;
    }
    int w_off[NUW];
#pragma unroll
    for (int u = 0; u < NUW; ++u) {
        const int row = ((wave + 4 * u) % NIW) * 16 + (lane >> 2);
        const int tl = row / NT, r = row - tl * NT;
        const int n = n0 + chan_of_row_v<NFRAG>(r >> 4, r & 15);
        w_off[u] = ((tl * p.N + n) * p.K) * ESZ + ((lane & 3) ^ swz(row)) * 16;
    }
    const char* const zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 3) * 16;

    int q_img, q_ty, q_tx, q_kc = 0, q_chunk = 0;
    {
        const int tiles_img = ntx * nty;
        q_img = tile0 / tiles_img;
        const int r = tile0 - q_img * tiles_img;
        q_ty = r / ntx; q_tx = r - q_ty * ntx;
    }
    int c_img = q_img, c_ty = q_ty, c_tx = q_tx;
    const int img_wg = q_img;

    auto issue_next = [&]() {
        if (q_chunk < nchunk) {
            const int iy0 = q_ty << 4, ix0 = q_tx << 4;             // gather offsets start at 0
            const bool interior = iy0 + PD <= H && ix0 + PD <= W;
            const bool first = q_kc < nkc0;
            const char* pb = first ? p.in0 + (((long long)q_img * H + iy0) * W + ix0) * spb0 + (p.in0_coff + q_kc * KCE) * ESZ
                                   : p.in1 + (((long long)q_img * H + iy0) * W + ix0) * spb1 + (p.in1_coff + (q_kc - nkc0) * KCE) * ESZ;
            char* dst = smP + (q_chunk & 1) * PBYTES;
#pragma unroll
            for (int u = 0; u < NUP; ++u) {
                const int i = (wave + 4 * u) % NIP;
                const int off = first ? d_off0[u] : d_off1[u];
                bool ok = d_yx[u] >= 0;
                if (!interior) ok = ok && iy0 + (d_yx[u] >> 16) < H && ix0 + (d_yx[u] & 0xffff) < W;
                dma16(ok ? pb + off : zero, dst + i * 1024);
            }
            const char* wb = p.w + (size_t)q_kc * KCE * ESZ;
            char* wdst = smW + (q_chunk & 1) * WBYTES;
#pragma unroll
            for (int u = 0; u < NUW; ++u) {
                const int i = (wave + 4 * u) % NIW;
                dma16(wb + w_off[u], wdst + i * 1024);
            }
        }
        ++q_chunk;
        if (++q_kc == nkc) { q_kc = 0; if (++q_tx == ntx) { q_tx = 0; if (++q_ty == nty) { q_ty = 0; ++q_img; } } }
    };

    int bx[2];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) bx[dx] = ((wave * 4) * PD + lr + dx) * 64 + ((g ^ swz(lr + dx)) << 4);
    const int w_lane = lr * 64 + ((g ^ swz(lr)) << 4);
    const int rel0 = (wave * 4 * 2) * p.od[2] + lr * 2;
    const int rel_step = 2 * p.od[2];

    const bool do_stats = p.stats != nullptr;
    float s1[NFRAG][4], s2[NFRAG][4];
#pragma unroll
    for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[nf][j] = 0.f; s2[nf][j] = 0.f; }

    issue_next();
    v4f acc[4][4][NFRAG];                          // [phase][base row of the wave][channel fragment]
    int kc = 0;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        const int poff = (chunk & 1) * PBYTES;
        const char* wbuf = smW + (chunk & 1) * WBYTES;
        if (kc == 0) {
#pragma unroll
            for (int ph = 0; ph < 4; ++ph)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                    for (int nf = 0; nf < NFRAG; ++nf) acc[ph][mf][nf] = v4f{0.f, 0.f, 0.f, 0.f};
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();              // this chunk has landed everywhere; the other buffers are free
        asm volatile("" ::: "memory");
        uint4 B[5][2];
#pragma unroll
        for (int r = 0; r < 5; ++r)
#pragma unroll
            for (int dx = 0; dx < 2; ++dx) B[r][dx] = *reinterpret_cast<const uint4*>(smem + poff + bx[dx] + r * PD * 64);
        uint4 afr[2][NFRAG];
#pragma unroll
        for (int nf = 0; nf < NFRAG; ++nf) afr[0][nf] = *reinterpret_cast<const uint4*>(wbuf + (nf * 16) * 64 + w_lane);
        issue_next();
#pragma unroll
        for (int t = 0; t < NTAP; ++t) {
            if (t + 1 < NTAP) {
#pragma unroll
                for (int nf = 0; nf < NFRAG; ++nf)
                    afr[(t + 1) & 1][nf] = *reinterpret_cast<const uint4*>(wbuf + ((t + 1) * NT + nf * 16) * 64 + w_lane);
            }
            const int ph = ph_of_tap(t), dy = ph_dy(t), dx = ph_dx(t);
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int nf = 0; nf < NFRAG; ++nf) acc[ph][mf][nf] = mma16v<TR>(afr[t & 1][nf], B[mf + dy][dx], acc[ph][mf][nf]);
        }
        if (++kc == nkc) {
            kc = 0;
            const int by0 = c_ty << 4, bx0 = c_tx << 4;
#pragma unroll
            for (int ph = 0; ph < 4; ++ph) {
                const int base_pix = (c_img * p.od[1] + by0 * 2 + (ph >> 1)) * p.od[2] + bx0 * 2 + (ph & 1);
                tile_epilogue<TR, NFRAG, ADD ? 2 : 1>(p, acc[ph], base_pix, rel0, rel_step, by0, bx0, wave, lr, g, n0, do_stats, s1, s2);
            }
            if (++c_tx == ntx) { c_tx = 0; if (++c_ty == nty) { c_ty = 0; ++c_img; } }
        }
    }
    if (do_stats) flush_stats<NFRAG>(p, smem, s1, s2, wave, lr, g, n0, img_wg, slot);
}

// tiles per workgroup: a divisor of the tiles of one image, aiming at about four workgroups per CU
static int tiles_per_wg(const IgemmParams& p, int NT) {
    const int tiles_img = p.ntile[1] * p.ntile[2];
    const long long wgs1 = (long long)p.nb * tiles_img * (p.N / NT);
    int want = (int)(wgs1 / 1024);
    if (want > 16) want = 16;
    if (want < 1) want = 1;
    int tpw = 1;
    for (int t = 1; t <= want; ++t) if (tiles_img % t == 0) tpw = t;
    return tpw;
}

template <class TR, int NFRAG>
int launch_conv2d_phase_t(const IgemmParams& pin, hipStream_t st) {
    IgemmParams p = pin;
    if (!p.out1) p.out0_c = p.N;
    constexpr int NT = NFRAG * 16;
    const size_t lds = (size_t)2 * 19 * 1024 + (size_t)2 * 9 * NT * 64;
    p.tpw = tiles_per_wg(p, NT);
    auto kern = p.addend ? conv2d_phase_kernel<TR, NFRAG, true> : conv2d_phase_kernel<TR, NFRAG, false>;
    if (lds > 48 * 1024) {
        hipError_t e = mi_func_set_lds(reinterpret_cast<const void*>(kern), (int)lds);
        if (e != hipSuccess) { mi_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return -2; }
    }
    dim3 grid(p.nb * p.ntile[1] * p.ntile[2] / p.tpw, p.N / NT, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("conv2d phase launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

// ---- stride-2 gather form: stride-2 convolution k3 p1, and the data gradient of the k3 s2 transposed convolution ------
// Output pixel (y, x) gathers input pixels (2y - 1 + ky, 2x - 1 + kx).  A tap reads one of the four polyphase
// components of the input: component (ry, rx) = pixels (2i - ry, 2j - rx), and inside a component the gather has unit
// stride, so the fragment reads look like those of the unit-stride kernel.  One stage = one (tile, chunk, component):
// a 17 x 17 component patch with 1, 2, 2 or 4 taps (ky = 1 lives in ry = 0 at row offset 0; ky = 0 and 2 in ry = 1 at
// row offsets 0 and 1; the same along x).  Patches are double-buffered.  WRES: the weights of all chunks stay in LDS;
// otherwise (wide contractions, 64 output channels per workgroup) a stage's 1, 2, 2 or 4 tap tiles travel with its
// patch into the stage's half of a double buffer: one 4 KiB tap tile = four DMA wave-instructions, one per wave.
__host__ __device__ constexpr int s2_ntap(int c) { return c == 0 ? 1 : c == 3 ? 4 : 2; }
__host__ __device__ constexpr int s2_tap(int c, int i) {     // i-th tap of component c = (ry, rx), in (iy, ix) order
    const int ry = c >> 1, rx = c & 1;
    const int iy = rx ? i >> 1 : i, ix = rx ? i & 1 : 0;
    return (ry ? 2 * iy : 1) * 3 + (rx ? 2 * ix : 1);
}
template <class TR, int NFRAG, bool WRES>
__global__ __launch_bounds__(256, 2) void conv2d_s2_kernel(const IgemmParams p) {
    signal_start(p.start_flag, p.start_val);
    constexpr int ESZ = TR::ESZ;
    constexpr int KCE = 64 / ESZ;
    constexpr int NT = NFRAG * 16;
    constexpr int PD = 17, PP = PD * PD;
    constexpr int NIP = (PP + 15) / 16;
    constexpr int PBYTES = NIP * 1024;
    constexpr int NUP = (NIP + 3) / 4;
    constexpr int NTAP = 9;
    constexpr int WROWS = NTAP * NT;
    constexpr int NIW = WROWS / 16;
    constexpr int NUW = (NIW + 3) / 4;
    constexpr int WBYTES = WROWS * 64;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    static_assert(WRES || NFRAG == 4, "streamed weights: one DMA instruction per wave and tap tile");
    constexpr int WSTAGE = 4 * NT * 64;            // streamed weights: room for the four tap tiles of component (1, 1)
    char* const smP = smem;
    char* const smW = smem + 2 * PBYTES;           // WRES: [chunk][tap][n] rows; else two stage buffers of [tap of the stage][n] rows

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    const int H = p.id[1], W = p.id[2];
    const int ntx = p.ntile[2], nty = p.ntile[1];
    const int tile0 = blockIdx.x * p.tpw;
    const int n0 = blockIdx.y * NT;
    const int slot = blockIdx.x % MI_NSLOT;
    const int nkc = p.K / KCE;
    const int nstage = p.tpw * nkc * 4;
    // streamed weights: this wave moves rows 16 wave .. 16 wave + 15 of every tap tile
    const int ws_off = WRES ? 0 : ((n0 + chan_of_row_v<NFRAG>(wave, lane >> 2)) * p.K) * ESZ + ((lane & 3) ^ swz(lane >> 2)) * 16;
    const int tap_bytes = p.N * p.K * ESZ;
    const int spb0 = p.in0_pitch * ESZ, spb1 = p.in1_pitch * ESZ;
    const int nkc0 = min(nkc, p.in0_c / KCE);

    int d_off0[NUP], d_off1[NUP], d_yx[NUP];       // component patch pixel (py, px) sits 2 py rows and 2 px columns from the origin
#pragma unroll
    for (int u = 0; u < NUP; ++u) {
        const int pix = ((wave + 4 * u) % NIP) * 16 + (lane >> 2);
        const int py = pix / PD, px = pix - py * PD;
        const int cb = ((lane & 3) ^ swz(px)) * 16;
        d_off0[u] = (2 * py * W + 2 * px) * spb0 + cb;
        d_off1[u] = (2 * py * W + 2 * px) * spb1 + cb;
        d_yx[u] = pix < PP ? ((py << 16) | px) : -1;
    }
    const char* const zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 3) * 16;

    int q_img, q_ty, q_tx, q_kc = 0, q_c = 0, q_stage = 0;
    {
        const int tiles_img = ntx * nty;
        q_img = tile0 / tiles_img;
        const int r = tile0 - q_img * tiles_img;
        q_ty = r / ntx; q_tx = r - q_ty * ntx;
    }
    int c_img = q_img, c_ty = q_ty, c_tx = q_tx;
    const int img_wg = q_img;

    auto issue_next = [&]() {
        if (q_stage < nstage) {
            const int ry = q_c >> 1, rx = q_c & 1;
            const int iy0 = (q_ty << 5) - ry, ix0 = (q_tx << 5) - rx;
            const bool first = q_kc < nkc0;
            const char* pb = first ? p.in0 + (((long long)q_img * H + iy0) * W + ix0) * spb0 + (p.in0_coff + q_kc * KCE) * ESZ
                                   : p.in1 + (((long long)q_img * H + iy0) * W + ix0) * spb1 + (p.in1_coff + (q_kc - nkc0) * KCE) * ESZ;
            char* dst = smP + (q_stage & 1) * PBYTES;
#pragma unroll
            for (int u = 0; u < NUP; ++u) {
                const int i = (wave + 4 * u) % NIP;
                const int off = first ? d_off0[u] : d_off1[u];
                const int iy = iy0 + 2 * (d_yx[u] >> 16), ix = ix0 + 2 * (d_yx[u] & 0xffff);
                const bool ok = d_yx[u] >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                dma16(ok ? pb + off : zero, dst + i * 1024);
            }
            if (!WRES) {
                const char* wb = p.w + (size_t)q_kc * KCE * ESZ + ws_off;
                char* wd = smW + (q_stage & 1) * WSTAGE + wave * 1024;
                // tap tiles of component q_c in fragment order; the counts 1, 2, 2, 4 are wave-uniform
                if (q_c == 0) {
                    dma16(wb + (size_t)s2_tap(0, 0) * tap_bytes, wd);
                } else if (q_c == 1) {
                    dma16(wb + (size_t)s2_tap(1, 0) * tap_bytes, wd); dma16(wb + (size_t)s2_tap(1, 1) * tap_bytes, wd + NT * 64);
                } else if (q_c == 2) {
                    dma16(wb + (size_t)s2_tap(2, 0) * tap_bytes, wd); dma16(wb + (size_t)s2_tap(2, 1) * tap_bytes, wd + NT * 64);
                } else {
#pragma unroll
                    for (int i = 0; i < 4; ++i) dma16(wb + (size_t)s2_tap(3, i) * tap_bytes, wd + i * NT * 64);
                }
            }
        }
        ++q_stage;
        if (++q_c == 4) {
            q_c = 0;
            if (++q_kc == nkc) { q_kc = 0; if (++q_tx == ntx) { q_tx = 0; if (++q_ty == nty) { q_ty = 0; ++q_img; } } }
        }
    };

    // resident weights: every chunk's [tap][n] rows
    for (int c = 0; WRES && c < nkc; ++c) {
#pragma unroll
        for (int u = 0; u < NUW; ++u) {
            const int i = (wave + 4 * u) % NIW;
            const int row = i * 16 + (lane >> 2);
            const int tl = row / NT, r = row - tl * NT;
            const int n = n0 + chan_of_row_v<NFRAG>(r >> 4, r & 15);
            dma16(p.w + ((size_t)(tl * p.N + n) * p.K + (size_t)c * KCE) * ESZ + ((lane & 3) ^ swz(row)) * 16, smW + c * WBYTES + i * 1024);
        }
    }

    int bx[2];
#pragma unroll
    for (int dx = 0; dx < 2; ++dx) bx[dx] = ((wave * 4) * PD + lr + dx) * 64 + ((g ^ swz(lr + dx)) << 4);
    const int w_lane = lr * 64 + ((g ^ swz(lr)) << 4);
    const int rel0 = (wave * 4) * p.od[2] + lr;
    const int rel_step = p.od[2];

    const bool do_stats = p.stats != nullptr;
    float s1[NFRAG][4], s2[NFRAG][4];
#pragma unroll
    for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[nf][j] = 0.f; s2[nf][j] = 0.f; }

    issue_next();
    v4f acc[4][NFRAG];
    int kc = 0, stage = 0;
    const int nchunk = p.tpw * nkc;
    for (int chunk = 0; chunk < nchunk; ++chunk) {
        if (kc == 0) {
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = v4f{0.f, 0.f, 0.f, 0.f};
        }
        const char* wres = smW + kc * WBYTES;
#pragma unroll
        for (int c = 0; c < 4; ++c, ++stage) {
            const int ry = c >> 1, rx = c & 1;
            const char* wbuf = WRES ? wres : smW + (stage & 1) * WSTAGE;
            const int nrow = 4 + ry, ncol = 1 + rx;
            const int poff = (stage & 1) * PBYTES;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            uint4 B[5][2];
#pragma unroll
            for (int r = 0; r < 5; ++r)
#pragma unroll
                for (int dx = 0; dx < 2; ++dx)
                    if (r < nrow && dx < ncol) B[r][dx] = *reinterpret_cast<const uint4*>(smem + poff + bx[dx] + r * PD * 64);
            issue_next();
#pragma unroll
            for (int iy = 0; iy <= ry; ++iy)
#pragma unroll
                for (int ix = 0; ix <= rx; ++ix) {
                    // ry = 0: ky = 1 at row offset 0; ry = 1: ky = 0 at offset 0, ky = 2 at offset 1 (x alike)
                    const int ky = ry ? 2 * iy : 1, kx = rx ? 2 * ix : 1;
                    const int t = WRES ? ky * 3 + kx : (rx ? 2 * iy + ix : iy);     // streamed: position inside the stage's tiles
                    uint4 afr[NFRAG];
#pragma unroll
                    for (int nf = 0; nf < NFRAG; ++nf) afr[nf] = *reinterpret_cast<const uint4*>(wbuf + (t * NT + nf * 16) * 64 + w_lane);
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                        for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = mma16v<TR>(afr[nf], B[mf + iy][ix], acc[mf][nf]);
                }
        }
        if (++kc == nkc) {
            kc = 0;
            const int by0 = c_ty << 4, bx0 = c_tx << 4;
            const int base_pix = (c_img * p.od[1] + by0) * p.od[2] + bx0;
            if (++c_tx == ntx) { c_tx = 0; if (++c_ty == nty) { c_ty = 0; ++c_img; } }
            tile_epilogue<TR, NFRAG>(p, acc, base_pix, rel0, rel_step, by0, bx0, wave, lr, g, n0, do_stats, s1, s2);
        }
    }
    if (do_stats) flush_stats<NFRAG>(p, smem, s1, s2, wave, lr, g, n0, img_wg, slot);
}

template <class TR, int NFRAG, bool WRES>
int launch_conv2d_s2_t(const IgemmParams& pin, hipStream_t st) {
    IgemmParams p = pin;
    if (!p.out1) p.out0_c = p.N;
    constexpr int NT = NFRAG * 16;
    const int nkc = p.K / (64 / TR::ESZ);
    const size_t lds = (size_t)2 * 19 * 1024 + (WRES ? (size_t)nkc * 9 * NT * 64 : (size_t)2 * 4 * NT * 64);
    p.tpw = tiles_per_wg(p, NT);
    auto kern = conv2d_s2_kernel<TR, NFRAG, WRES>;
    if (lds > 48 * 1024) {
        hipError_t e = mi_func_set_lds(reinterpret_cast<const void*>(kern), (int)lds);
        if (e != hipSuccess) { mi_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return -2; }
    }
    dim3 grid(p.nb * p.ntile[1] * p.ntile[2] / p.tpw, p.N / NT, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("conv2d s2 launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

// ---- 16 contraction channels (bf16): 32-byte pixel rows, two taps per MFMA contraction step ---------------------------
// With K = 16 a 16x16x32 MFMA would run half empty.  Here contraction slots 0..15 carry one tap and 16..31 the next:
// lanes 0..31 (g = 0, 1) read tap A's pixel and weight row, lanes 32..63 tap B's, so the nine taps take five steps
// (the ninth pairs with an all-zero weight block, and its partner lanes read that zero block as their "pixels").
// Rows are unpadded and unpermuted: 32-byte rows are conflict-free for these reads as they are.  One stage = one tile
// (the whole contraction fits one step set); patches double-buffered, the weight image (10 tap blocks) resident.
__host__ __device__ constexpr int k16_tap(int step, int hi) { return 2 * step + hi; }          // 9 = the zero block

template <int NFRAG>
__global__ __launch_bounds__(256, 2) void conv2d_k16_kernel(const IgemmParams p) {
    signal_start(p.start_flag, p.start_val);
    using TR = BF16;
    constexpr int ESZ = 2;
    constexpr int NT = NFRAG * 16;
    constexpr int PD = 18, PP = PD * PD;
    constexpr int NIP = (PP + 31) / 32;            // 11 DMA wave-instructions per patch (32 pixels each)
    constexpr int PBYTES = NIP * 1024;
    constexpr int NUP = (NIP + 3) / 4;
    constexpr int WROWS = 10 * NT;
    constexpr int NIW = WROWS / 32;
    constexpr int NUW = (NIW + 3) / 4;
    constexpr int NSTEP = 5;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const smP = smem;
    char* const smW = smem + 2 * PBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    const int H = p.id[1], W = p.id[2];
    const int ntx = p.ntile[2], nty = p.ntile[1];
    const int tile0 = blockIdx.x * p.tpw;
    const int n0 = blockIdx.y * NT;
    const int slot = blockIdx.x % MI_NSLOT;
    const int spb = p.in0_pitch * ESZ;

    int d_off[NUP], d_yx[NUP];
#pragma unroll
    for (int u = 0; u < NUP; ++u) {
        const int pix = ((wave + 4 * u) % NIP) * 32 + (lane >> 1);
        const int py = pix / PD, px = pix - py * PD;
        d_off[u] = (py * W + px) * spb + (lane & 1) * 16;
        d_yx[u] = pix < PP ? ((py << 16) | px) : -1;
    }
    const char* const zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 1) * 16;

    int q_img, q_ty, q_tx, q_tile = 0;
    {
        const int tiles_img = ntx * nty;
        q_img = tile0 / tiles_img;
        const int r = tile0 - q_img * tiles_img;
        q_ty = r / ntx; q_tx = r - q_ty * ntx;
    }
    int c_img = q_img, c_ty = q_ty, c_tx = q_tx;
    const int img_wg = q_img;

    auto issue_next = [&]() {
        if (q_tile < p.tpw) {
            const int iy0 = (q_ty << 4) + p.omin[1], ix0 = (q_tx << 4) + p.omin[2];
            const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + PD <= H && ix0 + PD <= W;
            const char* pb = p.in0 + (((long long)q_img * H + iy0) * W + ix0) * spb + p.in0_coff * ESZ;
            char* dst = smP + (q_tile & 1) * PBYTES;
#pragma unroll
            for (int u = 0; u < NUP; ++u) {
                const int i = (wave + 4 * u) % NIP;
                bool ok = d_yx[u] >= 0;
                if (!interior) ok = ok && (unsigned)(iy0 + (d_yx[u] >> 16)) < (unsigned)H && (unsigned)(ix0 + (d_yx[u] & 0xffff)) < (unsigned)W;
                dma16(ok ? pb + d_off[u] : zero, dst + i * 1024);
            }
        }
        ++q_tile;
        if (++q_tx == ntx) { q_tx = 0; if (++q_ty == nty) { q_ty = 0; ++q_img; } }
    };

    // resident weights: rows (tap, n) of 32 bytes, tap 9 all zero
#pragma unroll
    for (int u = 0; u < NUW; ++u) {
        const int i = (wave + 4 * u) % NIW;
        const int row = i * 32 + (lane >> 1);
        const int tl = row / NT, r = row - tl * NT;
        const int n = n0 + chan_of_row_v<NFRAG>(r >> 4, r & 15);
        dma16(tl < 9 ? p.w + ((size_t)(tl * p.N + n) * 16) * ESZ + (lane & 1) * 16 : zero, smW + i * 1024);
    }

    // per-lane fragment offsets of the five steps: the lane's tap is 2 * step + (g >> 1)
    int bq[NSTEP], aw[NSTEP];
#pragma unroll
    for (int st = 0; st < NSTEP; ++st) {
        const int t = k16_tap(st, g >> 1);
        aw[st] = (t * NT + lr) * 32 + (g & 1) * 16;
        if (t < 9) bq[st] = ((wave * 4 + t / 3) * PD + lr + t % 3) * 32 + (g & 1) * 16;
        else bq[st] = -1;                          // reads the zero weight block instead of a patch
    }
    const int zrow = 2 * PBYTES + (9 * NT + lr) * 32 + (g & 1) * 16;   // smem offset of a zero row
    const int rel0 = (wave * 4) * p.od[2] + lr;
    const int rel_step = p.od[2];

    const bool do_stats = p.stats != nullptr;
    float s1[NFRAG][4], s2[NFRAG][4];
#pragma unroll
    for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[nf][j] = 0.f; s2[nf][j] = 0.f; }

    issue_next();
    for (int tile = 0; tile < p.tpw; ++tile) {
        const int poff = (tile & 1) * PBYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        uint4 B[NSTEP][4];
#pragma unroll
        for (int st = 0; st < NSTEP; ++st)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
                B[st][mf] = *reinterpret_cast<const uint4*>(smem + (bq[st] >= 0 ? poff + bq[st] + mf * PD * 32 : zrow));
        issue_next();
        v4f acc[4][NFRAG];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int st = 0; st < NSTEP; ++st) {
            uint4 afr[NFRAG];
#pragma unroll
            for (int nf = 0; nf < NFRAG; ++nf) afr[nf] = *reinterpret_cast<const uint4*>(smW + aw[st] + nf * 16 * 32);
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = mma16v<TR>(afr[nf], B[st][mf], acc[mf][nf]);
        }
        const int by0 = c_ty << 4, bx0 = c_tx << 4;
        const int base_pix = (c_img * p.od[1] + by0) * p.od[2] + bx0;
        if (++c_tx == ntx) { c_tx = 0; if (++c_ty == nty) { c_ty = 0; ++c_img; } }
        tile_epilogue<TR, NFRAG>(p, acc, base_pix, rel0, rel_step, by0, bx0, wave, lr, g, n0, do_stats, s1, s2);
    }
    if (do_stats) flush_stats<NFRAG>(p, smem, s1, s2, wave, lr, g, n0, img_wg, slot);
}

template <int NFRAG>
int launch_conv2d_k16_t(const IgemmParams& pin, hipStream_t st) {
    IgemmParams p = pin;
    if (!p.out1) p.out0_c = p.N;
    constexpr int NT = NFRAG * 16;
    const size_t lds = (size_t)2 * 11 * 1024 + (size_t)10 * NT * 32;
    p.tpw = tiles_per_wg(p, NT);
    auto kern = conv2d_k16_kernel<NFRAG>;
    dim3 grid(p.nb * p.ntile[1] * p.ntile[2] / p.tpw, p.N / NT, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("conv2d k16 launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

// ---- stride-2 gather with 16 contraction channels (bf16): the component stages of conv2d_s2_kernel on 32-byte rows,
// two taps per MFMA step as in conv2d_k16_kernel.  Steps: component (0,0): tap (1,1) with the zero block; (0,1): taps
// (1,0) (1,2); (1,0): taps (0,1) (2,1); (1,1): taps (0,0) (0,2), then (2,0) (2,2).
__host__ __device__ constexpr int s2k16_comp(int step) { return step < 3 ? step : 3; }
__host__ __device__ constexpr int s2k16_tap(int step, int hi) {           // weight slot (ky * 3 + kx), 9 = zero block
    return step == 0 ? (hi ? 9 : 4) : step == 1 ? (hi ? 5 : 3) : step == 2 ? (hi ? 7 : 1) : step == 3 ? (hi ? 2 : 0) : (hi ? 8 : 6);
}
__host__ __device__ constexpr int s2k16_iy(int step, int hi) { return step == 2 ? hi : step == 4 ? 1 : 0; }   // row offset in the component patch
__host__ __device__ constexpr int s2k16_ix(int step, int hi) { return (step == 1 || step >= 3) ? hi : 0; }

template <int NFRAG>
__global__ __launch_bounds__(256, 2) void conv2d_s2k16_kernel(const IgemmParams p) {
    signal_start(p.start_flag, p.start_val);
    using TR = BF16;
    constexpr int ESZ = 2;
    constexpr int NT = NFRAG * 16;
    constexpr int PD = 17, PP = PD * PD;
    constexpr int NIP = (PP + 31) / 32;            // 10
    constexpr int PBYTES = NIP * 1024;
    constexpr int NUP = (NIP + 3) / 4;
    constexpr int WROWS = 10 * NT;
    constexpr int NIW = WROWS / 32;
    constexpr int NUW = (NIW + 3) / 4;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const smP = smem;
    char* const smW = smem + 2 * PBYTES;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    const int H = p.id[1], W = p.id[2];
    const int ntx = p.ntile[2], nty = p.ntile[1];
    const int tile0 = blockIdx.x * p.tpw;
    const int n0 = blockIdx.y * NT;
    const int slot = blockIdx.x % MI_NSLOT;
    const int nstage = p.tpw * 4;
    const int spb = p.in0_pitch * ESZ;

    int d_off[NUP], d_yx[NUP];
#pragma unroll
    for (int u = 0; u < NUP; ++u) {
        const int pix = ((wave + 4 * u) % NIP) * 32 + (lane >> 1);
        const int py = pix / PD, px = pix - py * PD;
        d_off[u] = (2 * py * W + 2 * px) * spb + (lane & 1) * 16;
        d_yx[u] = pix < PP ? ((py << 16) | px) : -1;
    }
    const char* const zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 1) * 16;

    int q_img, q_ty, q_tx, q_c = 0, q_stage = 0;
    {
        const int tiles_img = ntx * nty;
        q_img = tile0 / tiles_img;
        const int r = tile0 - q_img * tiles_img;
        q_ty = r / ntx; q_tx = r - q_ty * ntx;
    }
    int c_img = q_img, c_ty = q_ty, c_tx = q_tx;
    const int img_wg = q_img;

    auto issue_next = [&]() {
        if (q_stage < nstage) {
            const int ry = q_c >> 1, rx = q_c & 1;
            const int iy0 = (q_ty << 5) - ry, ix0 = (q_tx << 5) - rx;
            const char* pb = p.in0 + (((long long)q_img * H + iy0) * W + ix0) * spb + p.in0_coff * ESZ;
            char* dst = smP + (q_stage & 1) * PBYTES;
#pragma unroll
            for (int u = 0; u < NUP; ++u) {
                const int i = (wave + 4 * u) % NIP;
                const int iy = iy0 + 2 * (d_yx[u] >> 16), ix = ix0 + 2 * (d_yx[u] & 0xffff);
                const bool ok = d_yx[u] >= 0 && (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                dma16(ok ? pb + d_off[u] : zero, dst + i * 1024);
            }
        }
        ++q_stage;
        if (++q_c == 4) { q_c = 0; if (++q_tx == ntx) { q_tx = 0; if (++q_ty == nty) { q_ty = 0; ++q_img; } } }
    };

#pragma unroll
    for (int u = 0; u < NUW; ++u) {
        const int i = (wave + 4 * u) % NIW;
        const int row = i * 32 + (lane >> 1);
        const int tl = row / NT, r = row - tl * NT;
        const int n = n0 + chan_of_row_v<NFRAG>(r >> 4, r & 15);
        dma16(tl < 9 ? p.w + ((size_t)(tl * p.N + n) * 16) * ESZ + (lane & 1) * 16 : zero, smW + i * 1024);
    }

    int bq[5], aw[5];
#pragma unroll
    for (int st = 0; st < 5; ++st) {
        const int hi = g >> 1;
        const int t = hi ? s2k16_tap(st, 1) : s2k16_tap(st, 0);
        const int iy = hi ? s2k16_iy(st, 1) : s2k16_iy(st, 0), ix = hi ? s2k16_ix(st, 1) : s2k16_ix(st, 0);
        aw[st] = (t * NT + lr) * 32 + (g & 1) * 16;
        bq[st] = t < 9 ? ((wave * 4 + iy) * PD + lr + ix) * 32 + (g & 1) * 16 : -1;
    }
    const int zrow = 2 * PBYTES + (9 * NT + lr) * 32 + (g & 1) * 16;
    const int rel0 = (wave * 4) * p.od[2] + lr;
    const int rel_step = p.od[2];

    const bool do_stats = p.stats != nullptr;
    float s1[NFRAG][4], s2[NFRAG][4];
#pragma unroll
    for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[nf][j] = 0.f; s2[nf][j] = 0.f; }

    issue_next();
    int stage = 0;
    for (int tile = 0; tile < p.tpw; ++tile) {
        v4f acc[4][NFRAG];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
            for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int c = 0; c < 4; ++c, ++stage) {
            const int poff = (stage & 1) * PBYTES;
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            const int st0 = c, nst = c == 3 ? 2 : 1;     // steps of this component: c, and 4 after 3
            uint4 B[2][4];
#pragma unroll
            for (int k = 0; k < 2; ++k)
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
                    if (k < nst) B[k][mf] = *reinterpret_cast<const uint4*>(smem + (bq[st0 + k] >= 0 ? poff + bq[st0 + k] + mf * PD * 32 : zrow));
            issue_next();
#pragma unroll
            for (int k = 0; k < 2; ++k) {
                if (k < nst) {
                    uint4 afr[NFRAG];
#pragma unroll
                    for (int nf = 0; nf < NFRAG; ++nf) afr[nf] = *reinterpret_cast<const uint4*>(smW + aw[st0 + k] + nf * 16 * 32);
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                        for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = mma16v<TR>(afr[nf], B[k][mf], acc[mf][nf]);
                }
            }
        }
        const int by0 = c_ty << 4, bx0 = c_tx << 4;
        const int base_pix = (c_img * p.od[1] + by0) * p.od[2] + bx0;
        if (++c_tx == ntx) { c_tx = 0; if (++c_ty == nty) { c_ty = 0; ++c_img; } }
        tile_epilogue<TR, NFRAG>(p, acc, base_pix, rel0, rel_step, by0, bx0, wave, lr, g, n0, do_stats, s1, s2);
    }
    if (do_stats) flush_stats<NFRAG>(p, smem, s1, s2, wave, lr, g, n0, img_wg, slot);
}

template <int NFRAG>
int launch_conv2d_s2k16_t(const IgemmParams& pin, hipStream_t st) {
    IgemmParams p = pin;
    if (!p.out1) p.out0_c = p.N;
    constexpr int NT = NFRAG * 16;
    const size_t lds = (size_t)2 * 10 * 1024 + (size_t)10 * NT * 32;
    p.tpw = tiles_per_wg(p, NT);
    auto kern = conv2d_s2k16_kernel<NFRAG>;
    dim3 grid(p.nb * p.ntile[1] * p.ntile[2] / p.tpw, p.N / NT, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("conv2d s2k16 launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

// ---- 3-D, 16 contraction channels (bf16), 3 x 3 x 3 taps, unit stride ------------------------------------------------
// Tile = 4 (z, one plane per wave) x 4 (y, one row per accumulator) x 16 (x) base pixels on a 6 x 6 x 18 patch of
// 32-byte pixel rows.  Two taps share an MFMA contraction step as in conv2d_k16_kernel, paired so that a pixel
// fragment depends on the patch row y' = y + dy only, not on (y, dy).  Per patch row there are five fragments,
//   kind dz = 0..2: pixels (z + dz, y', x | x + 1)       taps (dz, dy, 0) | (dz, dy, 1)
//   kind 3        : pixels (z | z + 1, y', x + 2)        taps (0, dy, 2)  | (1, dy, 2)
//   kind 4        : pixel  (z + 2, y', x + 2)            tap  (2, dy, 2)  | zero weights
// and every fragment read from LDS feeds up to three accumulator rows (y = y' - dy): 30 fragment reads for 60 MFMAs
// per wave and tile, where a tap-major loop takes 70 reads for 56.  The 15 weight fragments of a lane stay in
// registers for the life of the workgroup; patches are double-buffered over the workgroup's tiles.  A generic tile
// of this size spends ten times the cycles of its MFMAs on per-tile address set-up: here that is six DMA requests
// per wave and tile with precomputed offsets.
constexpr int C3_PDY = 6, C3_PDX = 18, C3_PP = 6 * C3_PDY * C3_PDX;      // 648 patch pixels
constexpr int C3_NIP = (C3_PP + 31) / 32;       // 21 DMA wave-instructions per patch (32 pixels each)
constexpr int C3_PBYTES = C3_NIP * 1024;
constexpr int C3_NUP = (C3_NIP + 3) / 4;

__global__ __launch_bounds__(256, 2) void conv3d_k16_kernel(const IgemmParams p) {
    signal_start(p.start_flag, p.start_val);
    using TR = BF16;
    constexpr int ESZ = 2;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4, hi = g >> 1, half = g & 1;
    const int D = p.id[0], H = p.id[1], W = p.id[2];
    const int ntx = p.ntile[2], nty = p.ntile[1], ntz = p.ntile[0];
    const int tile0 = blockIdx.x * p.tpw;
    const int n0 = blockIdx.y * 16;
    const int slot = blockIdx.x % MI_NSLOT;
    const int spb = p.in0_pitch * ESZ;

    int d_off[C3_NUP], d_zyx[C3_NUP];
#pragma unroll
    for (int u = 0; u < C3_NUP; ++u) {
        const int pix = ((wave + 4 * u) % C3_NIP) * 32 + (lane >> 1);
        const int pz = pix / (C3_PDY * C3_PDX), r = pix - pz * (C3_PDY * C3_PDX);
        const int py = r / C3_PDX, px = r - py * C3_PDX;
        d_off[u] = ((pz * H + py) * W + px) * spb + (lane & 1) * 16;
        d_zyx[u] = pix < C3_PP ? ((pz << 20) | (py << 10) | px) : -1;
    }
    const char* const zero = reinterpret_cast<const char*>(g_zero_page) + (lane & 1) * 16;

    int q_img, q_tz, q_ty, q_tx, q_tile = 0;
    {
        const int tiles_img = ntx * nty * ntz;
        q_img = tile0 / tiles_img;
        int r = tile0 - q_img * tiles_img;
        q_tz = r / (ntx * nty); r -= q_tz * ntx * nty;
        q_ty = r / ntx; q_tx = r - q_ty * ntx;
    }
    int c_img = q_img, c_tz = q_tz, c_ty = q_ty, c_tx = q_tx;
    const int img_wg = q_img;

    auto issue_next = [&]() {
        if (q_tile < p.tpw) {
            const int iz0 = (q_tz << 2) - 1, iy0 = (q_ty << 2) - 1, ix0 = (q_tx << 4) - 1;
            const bool interior = iz0 >= 0 && iy0 >= 0 && ix0 >= 0 && iz0 + 6 <= D && iy0 + C3_PDY <= H && ix0 + C3_PDX <= W;
            const char* pb = p.in0 + ((((long long)q_img * D + iz0) * H + iy0) * W + ix0) * spb + p.in0_coff * ESZ;
            char* dst = smem + (q_tile & 1) * C3_PBYTES;
#pragma unroll
            for (int u = 0; u < C3_NUP; ++u) {
                const int i = (wave + 4 * u) % C3_NIP;
                bool ok = d_zyx[u] >= 0;
                if (!interior)
                    ok = ok && (unsigned)(iz0 + (d_zyx[u] >> 20)) < (unsigned)D && (unsigned)(iy0 + ((d_zyx[u] >> 10) & 1023)) < (unsigned)H &&
                         (unsigned)(ix0 + (d_zyx[u] & 1023)) < (unsigned)W;
                dma16(ok ? pb + d_off[u] : zero, dst + i * 1024);
            }
        }
        ++q_tile;
        if (++q_tx == ntx) { q_tx = 0; if (++q_ty == nty) { q_ty = 0; if (++q_tz == ntz) { q_tz = 0; ++q_img; } } }
    };
    issue_next();

    // the lane's weight fragments: row n0 + lr of tap (dz, dy, dx), contraction half `half`
    uint4 wf[3][5];
#pragma unroll
    for (int dy = 0; dy < 3; ++dy)
#pragma unroll
        for (int k = 0; k < 5; ++k) {
            const int t = k < 3 ? k * 9 + dy * 3 + hi : k == 3 ? hi * 9 + dy * 3 + 2 : (hi ? -1 : 18 + dy * 3 + 2);
            wf[dy][k] = t >= 0 ? *reinterpret_cast<const uint4*>(p.w + ((size_t)(t * p.N + n0 + lr) * 16 + half * 8) * ESZ) : uint4{0u, 0u, 0u, 0u};
        }
    // LDS byte offsets of the five fragment kinds at patch row 0 (a patch row is C3_PDX * 32 bytes further on)
    int fb[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        const int pz = k < 3 ? wave + k : k == 3 ? wave + hi : wave + 2;
        const int px = k < 3 ? lr + hi : lr + 2;
        fb[k] = ((pz * C3_PDY) * C3_PDX + px) * 32 + half * 16;
    }

    const bool do_stats = p.stats != nullptr;
    float s1[1][4], s2[1][4];
#pragma unroll
    for (int j = 0; j < 4; ++j) { s1[0][j] = 0.f; s2[0][j] = 0.f; }
    const int plane = p.od[1] * p.od[2];
    const int rel0 = wave * plane + lr;

    for (int tile = 0; tile < p.tpw; ++tile) {
        const char* pbuf = smem + (tile & 1) * C3_PBYTES;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        issue_next();       // into the other buffer: every wave has left the tile that read it
        v4f acc[4][1];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) acc[mf][0] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
        for (int yp = 0; yp < C3_PDY; ++yp) {
            uint4 F[5];
#pragma unroll
            for (int k = 0; k < 5; ++k) F[k] = *reinterpret_cast<const uint4*>(pbuf + fb[k] + yp * C3_PDX * 32);
#pragma unroll
            for (int dy = 0; dy < 3; ++dy) {
                const int mf = yp - dy;
                if (mf >= 0 && mf < 4) {
#pragma unroll
                    for (int k = 0; k < 5; ++k) acc[mf][0] = mma16v<TR>(wf[dy][k], F[k], acc[mf][0]);
                }
            }
        }
        const int bz0 = c_tz << 2, by0 = c_ty << 2, bx0 = c_tx << 4;
        const int base_pix = ((c_img * p.od[0] + bz0) * p.od[1] + by0) * p.od[2] + bx0;
        int rows = 0;
        if (bz0 + wave < p.bd[0]) {
#pragma unroll
            for (int mf = 0; mf < 4; ++mf) rows |= (by0 + mf < p.bd[1]) << mf;
        }
        if (++c_tx == ntx) { c_tx = 0; if (++c_ty == nty) { c_ty = 0; if (++c_tz == ntz) { c_tz = 0; ++c_img; } } }
        tile_epilogue<TR, 1, 0, true>(p, acc, base_pix, rel0, p.od[2], rows, bx0 + lr < p.bd[2], wave, lr, g, n0, do_stats, s1, s2);
    }
    if (do_stats) flush_stats<1>(p, smem, s1, s2, wave, lr, g, n0, img_wg, slot);
}

int launch_conv3d_k16_t(const IgemmParams& pin, hipStream_t st) {
    IgemmParams p = pin;
    p.out0_c = p.N;
    p.ntile[0] = (p.bd[0] + 3) / 4; p.ntile[1] = (p.bd[1] + 3) / 4; p.ntile[2] = (p.bd[2] + 15) / 16;
    const int tiles_img = p.ntile[0] * p.ntile[1] * p.ntile[2];
    {
        // tiles per workgroup: a divisor of the tiles of one image (the statistics partials of a workgroup belong to
        // one image), aiming at about three workgroups per CU
        static const int div_env = getenv("MI_C3K16_WGS") ? atoi(getenv("MI_C3K16_WGS")) : 0;
        const long long wgs1 = (long long)p.nb * tiles_img * (p.N / 16);
        int want = (int)(wgs1 / (div_env > 0 ? div_env : 768));
        if (want > 16) want = 16;
        if (want < 1) want = 1;
        int tpw = 1;
        for (int t = 1; t <= want; ++t) if (tiles_img % t == 0) tpw = t;
        p.tpw = tpw;
    }
    dim3 grid(p.nb * tiles_img / p.tpw, p.N / 16, 1);
    hipLaunchKernelGGL(conv3d_k16_kernel, grid, dim3(256), (size_t)2 * C3_PBYTES, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("conv3d k16 launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

template <class TR, int NFRAG, int KS, bool WRES>
int launch_conv2d_t(const IgemmParams& pin, hipStream_t st) {
    IgemmParams p = pin;
    if (!p.out1) p.out0_c = p.N;
    constexpr int NT = NFRAG * 16, PD = KS == 3 ? 18 : 16, NIP = (PD * PD + 15) / 16, NTAP = KS * KS;
    const int nkc = p.K / (64 / TR::ESZ);
    const size_t lds = (size_t)2 * NIP * 1024 + (WRES ? (size_t)nkc * NTAP * NT * 64 : (size_t)3 * (NTAP / KS) * NT * 64);
    const int tiles_img = p.ntile[1] * p.ntile[2];
    {
        // tiles per workgroup: a divisor of the tiles of one image, aiming at about four workgroups per CU
        const long long wgs1 = (long long)p.nb * tiles_img * (p.N / NT);
        int want = (int)(wgs1 / 1024);
        if (want > 16) want = 16;
        if (want < 1) want = 1;
        int tpw = 1;
        for (int t = 1; t <= want; ++t) if (tiles_img % t == 0) tpw = t;
        p.tpw = tpw;
    }
    auto kern = conv2d_dma_kernel<TR, NFRAG, KS, WRES>;
    if (lds > 48 * 1024) {
        hipError_t e = mi_func_set_lds(reinterpret_cast<const void*>(kern), (int)lds);
        if (e != hipSuccess) { mi_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return -2; }
    }
    dim3 grid(p.nb * tiles_img / p.tpw, p.N / NT, 1);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("conv2d launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

}  // namespace

// Geometry this file serves: 2-D tensor, 16 x 16 base-pixel tile, unit gather stride, contraction channels in whole
// 64-byte chunks from either source, and either
//   1: one phase with 9 taps on an 18 x 18 patch or one tap on a 16 x 16 patch (output stride 1), or
//   2: the four stride-2 scatter phases with 1 + 2 + 2 + 4 taps on a 17 x 17 patch (output stride 2), or
//   3: gather stride 2 with 9 taps (four 17 x 17 polyphase patches), weights resident, or
//   4: bf16 with 16 contraction channels, 3 x 3, unit stride (two taps per MFMA step), or
//   5: the same with gather stride 2 (component stages).
// stride-2 gather: channel fragments per workgroup such that the weights of all chunks (9 taps) fit in 40 KiB
static int s2_nfrag(int dtype, const IgemmParams& p) {
    const int nkc = p.K / (64 / (dtype ? 2 : 4));
    for (int nf = 4; nf >= 1; nf >>= 1)
        if (p.N % (nf * 16) == 0 && (size_t)nkc * 9 * nf * 16 * 64 <= 40 * 1024) return nf;
    return 0;
}
// stride-2 gather with streamed weights: 64 output channels per workgroup where the resident form would have to
// take fewer (every channel block stages the same patches again)
static bool s2_streamed(int dtype, const IgemmParams& p) {
    static const bool off = getenv("MI_NO_CONV_S2_STREAM") != nullptr;
    // 64-channel workgroups only where they still fill the chip (8 slices of 256 x 256: 32 workgroups on the deepest
    // stride-2 layer, 0.764 -> 0.759 ms per step with the narrower resident form)
    const long long wgs = (long long)p.nb * p.ntile[1] * p.ntile[2] * (p.N / 64);
    return !off && p.N % 64 == 0 && s2_nfrag(dtype, p) < 4 && (wgs >= 256 || s2_nfrag(dtype, p) == 0);
}
static int conv2d_dma_kind(int dtype, const IgemmParams& p) {
    const int kce = 64 / (dtype ? 2 : 4);
    if (p.id[0] != 1 || p.tlog[0] != 0 || p.tlog[1] != 4 || p.tlog[2] != 4 || p.pd[0] != 1) return 0;
    if (p.ksplit > 1) return 0;
    {
        // per-lane DMA source offsets are 32-bit byte offsets from the patch origin: (row * W + column) * pixel pitch over
        // the rows of one patch (stride-2 forms step two pixels) must stay below 2^31 for the widest source
        const long long pitch = p.in0_c < p.K && p.in1_pitch > p.in0_pitch ? p.in1_pitch : p.in0_pitch;
        if ((long long)(p.pd[1] + 2) * p.id[2] * pitch * (dtype ? 2 : 4) >= (1ll << 31)) return 0;
    }
    if (dtype && p.K == 16 && p.in0_c >= 16 && p.N % 16 == 0 && p.sa == 1 && p.so == 1 && p.nphase == 1 && p.phase_start[1] == 9 &&
        p.pd[1] == 18 && p.pd[2] == 18 && p.id[1] < 32768 && p.id[2] < 32768) {
        // 16 bf16 contraction channels from one source, 3 x 3 taps in patch order
        for (int t = 0; t < 9; ++t)
            if (p.tap_off[t] != (t / 3) * 18 + t % 3) return 0;
        return 4;
    }
    if (dtype && p.K == 16 && p.in0_c >= 16 && p.N % 16 == 0 && p.sa == 2 && p.so == 1 && p.nphase == 1 && p.phase_start[1] == 9 &&
        p.pd[1] == 33 && p.pd[2] == 33 && p.omin[1] == -1 && p.omin[2] == -1 && p.id[1] < 32768 && p.id[2] < 32768) {
        for (int t = 0; t < 9; ++t)
            if (p.tap_off[t] != (t / 3) * 33 + t % 3) return 0;
        return 5;
    }
    if (p.K % kce || (p.in0_c < p.K && p.in0_c % kce) || p.N % 16) return 0;
    if ((long long)p.id[1] >= 32768 || (long long)p.id[2] >= 32768) return 0;
    if (p.sa == 2) {
        // stride-2 gather: 9 taps in kernel order on a 33 x 33 patch starting at offset -1; weights of all chunks resident
        if (p.so != 1 || p.nphase != 1 || p.phase_start[1] != 9 || p.pd[1] != 33 || p.pd[2] != 33) return 0;
        if (p.omin[1] != -1 || p.omin[2] != -1) return 0;
        for (int t = 0; t < 9; ++t)
            if (p.tap_off[t] != (t / 3) * 33 + t % 3) return 0;
        if (s2_nfrag(dtype, p) == 0 && !s2_streamed(dtype, p)) return 0;
        return 3;
    }
    if (p.sa != 1) return 0;
    if (p.so == 1 && p.nphase == 1) {
        const int ntaps = p.phase_start[1];
        const bool k3 = ntaps == 9 && p.pd[1] == 18 && p.pd[2] == 18, k1 = ntaps == 1 && p.pd[1] == 16 && p.pd[2] == 16;
        if (!k3 && !k1) return 0;
        for (int t = 0; t < ntaps; ++t)
            if (p.tap_off[t] != (k3 ? (t / 3) * 18 + t % 3 : 0)) return 0;      // taps in patch order (igemm_geometry sorts them)
        return 1;
    }
    if (p.so == 2 && p.nphase == 4 && p.pd[1] == 17 && p.pd[2] == 17 && p.phase_start[4] == 9) {
        if (p.omin[1] != 0 || p.omin[2] != 0) return 0;
        for (int ph = 0; ph < 4; ++ph)
            if (p.phase_off[ph][1] != (ph >> 1) || p.phase_off[ph][2] != (ph & 1)) return 0;
        for (int t = 0; t < 9; ++t) {
            if (t < p.phase_start[ph_of_tap(t)] || t >= p.phase_start[ph_of_tap(t) + 1]) return 0;
            if (p.tap_off[t] != ph_dy(t) * 17 + ph_dx(t)) return 0;
        }
        if (p.id[1] * 2 != p.od[1] || p.id[2] * 2 != p.od[2]) return 0;
        return 2;
    }
    return 0;
}
bool conv2d_dma_applicable(int dtype, const IgemmParams& p) {
    static const bool no_phase = getenv("MI_NO_CONV_PHASE") != nullptr;
    const int k = conv2d_dma_kind(dtype, p);
    static const bool no_s2 = getenv("MI_NO_CONV_S2") != nullptr;
    static const bool no_k16 = getenv("MI_NO_CONV_K16") != nullptr;
    return k == 1 || (k == 2 && !no_phase) || (k == 3 && !no_s2) || ((k == 4 || k == 5) && !no_k16);
}

// name of the kernel launch_conv2d_dma picks, as the step tables print it
void conv2d_dma_label(int dtype, const IgemmParams& p, char* buf, size_t n) {
    const char* tr = dtype ? "BF16" : "F32";
    const int kind = conv2d_dma_kind(dtype, p);
    if (kind == 5) snprintf(buf, n, "conv2d_s2k16_kernel<%d>", (p.N % 64 == 0) ? 4 : (p.N % 32 == 0) ? 2 : 1);
    else if (kind == 4) snprintf(buf, n, "conv2d_k16_kernel<%d>", (p.N % 64 == 0) ? 4 : (p.N % 32 == 0) ? 2 : 1);
    else if (kind == 3 && s2_streamed(dtype, p)) snprintf(buf, n, "conv2d_s2_kernel<%s,4,streamed>", tr);
    else if (kind == 3) snprintf(buf, n, "conv2d_s2_kernel<%s,%d>", tr, s2_nfrag(dtype, p));
    else if (kind == 2) snprintf(buf, n, "conv2d_phase_kernel<%s,%d>", tr, (p.N % 32 == 0) ? 2 : 1);
    else snprintf(buf, n, "conv2d_dma_kernel<%s,%d>", tr, (p.N % 64 == 0) ? 4 : (p.N % 32 == 0) ? 2 : 1);
}

int launch_conv2d_dma(int dtype, const IgemmParams& p, hipStream_t st) {
    if (conv2d_dma_kind(dtype, p) == 5) {
        if (p.N % 64 == 0) return launch_conv2d_s2k16_t<4>(p, st);
        if (p.N % 32 == 0) return launch_conv2d_s2k16_t<2>(p, st);
        return launch_conv2d_s2k16_t<1>(p, st);
    }
    if (conv2d_dma_kind(dtype, p) == 4) {
        if (p.N % 64 == 0) return launch_conv2d_k16_t<4>(p, st);
        if (p.N % 32 == 0) return launch_conv2d_k16_t<2>(p, st);
        return launch_conv2d_k16_t<1>(p, st);
    }
    if (conv2d_dma_kind(dtype, p) == 3) {
        const int nfs = s2_nfrag(dtype, p);
        if (s2_streamed(dtype, p)) return dtype ? launch_conv2d_s2_t<BF16, 4, false>(p, st) : launch_conv2d_s2_t<F32, 4, false>(p, st);
#define S2_CASE(TRT) do { if (nfs == 4) return launch_conv2d_s2_t<TRT, 4, true>(p, st); if (nfs == 2) return launch_conv2d_s2_t<TRT, 2, true>(p, st); return launch_conv2d_s2_t<TRT, 1, true>(p, st); } while (0)
        if (dtype) S2_CASE(BF16); else S2_CASE(F32);
#undef S2_CASE
    }
    if (conv2d_dma_kind(dtype, p) == 2) {
        const int nfp = (p.N % 32 == 0) ? 2 : 1;
        if (dtype) return nfp == 2 ? launch_conv2d_phase_t<BF16, 2>(p, st) : launch_conv2d_phase_t<BF16, 1>(p, st);
        return nfp == 2 ? launch_conv2d_phase_t<F32, 2>(p, st) : launch_conv2d_phase_t<F32, 1>(p, st);
    }
    const int ntaps = p.phase_start[1];
    const int nkc = p.K / (64 / (dtype ? 2 : 4));
    const int nf = (p.N % 64 == 0) ? 4 : (p.N % 32 == 0) ? 2 : 1;
    // resident weights when all chunks fit in 40 KiB next to the two patch buffers
    const bool wres = (size_t)nkc * ntaps * nf * 16 * 64 <= 40 * 1024;
#define CV_CASE(TRT, NF)                                                                       \
    if (nf == NF) {                                                                            \
        if (ntaps == 9) return wres ? launch_conv2d_t<TRT, NF, 3, true>(p, st) : launch_conv2d_t<TRT, NF, 3, false>(p, st); \
        return wres ? launch_conv2d_t<TRT, NF, 1, true>(p, st) : launch_conv2d_t<TRT, NF, 1, false>(p, st); \
    }
    if (dtype) { CV_CASE(BF16, 4) CV_CASE(BF16, 2) CV_CASE(BF16, 1) }
    else { CV_CASE(F32, 4) CV_CASE(F32, 2) CV_CASE(F32, 1) }
#undef CV_CASE
    mi_set_error("launch_conv2d_dma: no kernel variant");
    return -1;
}

// 3-D layers conv3d_k16_kernel serves: bf16, 16 contraction channels from one source, 27 taps in patch order, unit
// stride, one destination, no contraction split.
bool conv3d_k16_applicable(int dtype, const IgemmParams& p) {
    static const bool off = getenv("MI_NO_CONV3D_K16") != nullptr;
    if (off || !dtype || p.K != 16 || p.in0_c < 16 || p.N % 16 || p.N > 32 || p.sa != 1 || p.so != 1 || p.nphase != 1) return false;
    if (p.id[0] < 2 || p.phase_start[1] != 27 || p.out1 || p.ksplit > 1) return false;
    for (int d = 0; d < 3; ++d)
        if (p.omin[d] != -1 || p.id[d] != p.bd[d] || p.od[d] != p.bd[d]) return false;
    for (int t = 0; t < 27; ++t)
        if (p.tap_off[t] != ((t / 9) * p.pd[1] + (t / 3) % 3) * p.pd[2] + t % 3) return false;
    // 32-bit byte offsets inside a patch and 32-bit pixel indices
    if ((long long)8 * p.id[1] * p.id[2] * p.in0_pitch * 2 >= (1ll << 31)) return false;
    if ((long long)p.nb * p.id[0] * p.id[1] * p.id[2] >= (1ll << 31)) return false;
    return true;
}

int launch_conv3d_k16(const IgemmParams& p, hipStream_t st) { return launch_conv3d_k16_t(p, st); }
