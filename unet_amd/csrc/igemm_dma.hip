// Gather-GEMM convolution with asynchronous staging for ANY tile geometry (gfx950): the contract of igemm_kernel
// (kparams.h, IgemmParams: 256 base pixels per tile with run-time extents, run-time tap table, phases, contraction
// split) for unit gather stride.  It serves what the compile-time 2-D kernels of conv2d.hip do not: the unit-stride
// 3-D layers (3 x 3 x 3 and 1 x 1 x 1 convolutions and their data gradients) and the unit-stride 2-D layers that run
// under a contraction split.
//
// What differs from igemm_kernel:
//  * the input patch of a contraction chunk and the weight tile of a stage (a group of at most nine taps) go from
//    global memory straight into LDS (global_load ... lds, 16 bytes per lane, no registers in between).  A thread of
//    igemm_kernel can hold only a few 16-byte vectors in registers, so a patch of 600 pixels and a weight tile of 576
//    rows cost it seven dependent round trips per stage; here every load of a stage is issued back to back, and the
//    loads of stage s + 1 are in flight under the MFMA loop of stage s (two patch buffers, two weight buffers, one
//    barrier per stage).  The deep levels of a 3-D network, whose layers are a handful of such stages, are bound by
//    exactly these round trips;
//  * LDS rows are unpadded 64-byte (or 32-byte) pixel and weight rows.  With 64-byte rows the 16-byte piece c of row r
//    sits in slot c ^ (((r >> 2) & 1) << 1); the DMA writes LDS lane-linearly, so the permutation is applied to the
//    per-lane SOURCE address (for a lane it is the same in every wave-instruction) and again on the fragment reads;
//  * pixels outside the tensor, and the padding tap of an odd tap group with 32-byte rows, are read from a zero page.
#include "common.h"
#include "kparams.h"
#include "epilogue.h"

namespace {

static __device__ __align__(256) unsigned char g_zero_rows[256];

// 16 bytes per lane from global memory to LDS at lds_base + 16 * lane (lds_base wave-uniform).  Inline assembly on
// purpose: the compiler drains every LDS-DMA it knows of in front of the next LDS read, which would undo the prefetch.
// M0 holds the wave-uniform LDS destination and is restored.
__device__ __forceinline__ void dma16g(const char* gsrc, char* lds_base) {
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

__device__ __forceinline__ int swz_row(int row) { return ((row >> 2) & 1) << 1; }

constexpr int DMA_MAXUP = 11;   // patch wave-instructions per wave: at most 44 KiB of patch per buffer
constexpr int DMA_MAXUW = 9;    // weight wave-instructions per wave: at most 36 KiB of weights per buffer

template <class TR, int CPS, int NFRAG>
__global__ __launch_bounds__(256) void igemm_dma_kernel(const IgemmParams p) {
    signal_start(p.start_flag, p.start_val);
    constexpr int ESZ = TR::ESZ;
    constexpr int ROWB = CPS * 16;       // bytes of a pixel row / weight row in LDS = one contraction chunk
    constexpr int KCE = ROWB / ESZ;      // contraction elements per chunk
    constexpr int NT = NFRAG * 16;
    constexpr int RPI = 64 / CPS;        // LDS rows one DMA wave-instruction fills
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;

    int bid = blockIdx.x;
    const int txi = bid % p.ntile[2]; bid /= p.ntile[2];
    const int tyi = bid % p.ntile[1]; bid /= p.ntile[1];
    const int tzi = bid % p.ntile[0];
    const int img = bid / p.ntile[0];
    const int slot = blockIdx.x % MI_NSLOT;
    const int n0 = blockIdx.y * NT;
    const int ph = blockIdx.z % p.nphase;
    const int ksp = blockIdx.z / p.nphase;       // contraction split: which share of the chunks this workgroup takes
    const int bz0 = tzi << p.tlog[0], by0 = tyi << p.tlog[1], bx0 = txi << p.tlog[2];
    // stride-2 gathers stage one polyphase component of the gathered tensor per stage (unit stride inside it)
    const bool comp = p.comp_mode != 0;
    const int pdz = comp ? p.cpd[0] : p.pd[0], pdy = comp ? p.cpd[1] : p.pd[1], pdx = comp ? p.cpd[2] : p.pd[2];
    const int PP = pdz * pdy * pdx;
    const int pd12 = pdy * pdx;
    const int nip = (PP + RPI - 1) / RPI;        // DMA wave-instructions per patch
    const int pbytes = nip * 1024;
    const int tapg = comp ? 8 : p.tapg;          // taps per stage (component mode: the largest component)
    const int tapg_pad = CPS == 2 ? ((tapg + 1) & ~1) : tapg;
    const int niw = tapg_pad * NT / RPI;         // DMA wave-instructions per weight stage
    const int wbytes = niw * 1024;
    // a workgroup that takes a single contraction chunk stages one patch only: one patch buffer, more workgroups per CU
    const int nkc_wg = p.ksplit > 1 ? (((p.K + KCE - 1) / KCE + p.ksplit - 1) / p.ksplit) : (p.K + KCE - 1) / KCE;
    char* const smP = smem;                      // two patch buffers (one for a single chunk)
    char* const smW = smem + ((nkc_wg > 1 || comp) ? 2 : 1) * pbytes;         // two weight buffers
    const int t_begin = p.phase_start[ph], t_end = p.phase_start[ph + 1];

    // ---- per-lane DMA constants ------------------------------------------------------------------------------------
    // Wave-instruction i fills LDS rows i * RPI .. i * RPI + RPI - 1, CPS lanes per row; a lane's row number modulo 8
    // is the same in every instruction, and so is the source piece its slot receives.
    const int cb = CPS == 4 ? (((lane & 3) ^ swz_row(lane >> 2)) << 4) : ((lane & 1) << 4);
    const int nup = (nip - wave + 3) >> 2, nuw = (niw - wave + 3) >> 2;      // this wave's share (instructions wave, wave + 4, ...)
    int d_gp[DMA_MAXUP];                         // source pixel (index into the gathered tensor), -1: outside -> zero page
    {
        const int iz0 = bz0 * p.sa + p.omin[0], iy0 = by0 * p.sa + p.omin[1], ix0 = bx0 * p.sa + p.omin[2];
#pragma unroll
        for (int u = 0; u < DMA_MAXUP; ++u) {
            const int pix = (wave + 4 * u) * RPI + lane / CPS;
            const int pz = fast_div(pix, p.mg_pd12); const int r = pix - pz * pd12;
            const int py = fast_div(r, p.mg_pd2); const int px = r - py * pdx;
            if (comp) {          // component mode: the patch coordinates (10 bits each), the source pixel follows per stage
                d_gp[u] = pix < PP ? (pz << 20) | (py << 10) | px : -1;
                continue;
            }
            const int iz = iz0 + pz, iy = iy0 + py, ix = ix0 + px;
            const bool ok = pix < PP && (unsigned)iz < (unsigned)p.id[0] && (unsigned)iy < (unsigned)p.id[1] && (unsigned)ix < (unsigned)p.id[2];
            d_gp[u] = ok ? ((img * p.id[0] + iz) * p.id[1] + iy) * p.id[2] + ix : -1;
        }
    }
    unsigned w_d[DMA_MAXUW];                     // tap within the stage << 27 | byte offset of the row's piece inside a tap group
#pragma unroll
    for (int u = 0; u < DMA_MAXUW; ++u) {
        const int row = (wave + 4 * u) * RPI + lane / CPS;
        const int tl = row / NT, r = row - tl * NT;
        const int n = n0 + chan_of_row<NFRAG>(r >> 4, r & 15);
        w_d[u] = ((unsigned)tl << 27) | (unsigned)(((tl * p.N + n) * p.K) * ESZ + cb);
    }
    const char* const zero = reinterpret_cast<const char*>(g_zero_rows) + cb;

    auto issue_patch = [&](int kc, int c, int buf) {
        const char* src; int spb, scoff;
        if (kc < p.in0_c) { src = p.in0; spb = p.in0_pitch * ESZ; scoff = p.in0_coff + kc; }
        else { src = p.in1; spb = p.in1_pitch * ESZ; scoff = p.in1_coff + kc - p.in0_c; }
        const char* base = src + scoff * ESZ + cb;
        char* dst = smP + buf * pbytes + wave * 1024;
        if (comp) {
            // component c = (rz, ry, rx): patch pixel (pz, py, px) is input pixel 2 (tile origin + p) - r
            const int oz = 2 * bz0 - ((c >> 2) & 1), oy = 2 * by0 - ((c >> 1) & 1), ox = 2 * bx0 - (c & 1);
#pragma unroll
            for (int u = 0; u < DMA_MAXUP; ++u)
                if (u < nup) {
                    const int q = d_gp[u];
                    const int iz = oz + 2 * (q >> 20), iy = oy + 2 * ((q >> 10) & 1023), ix = ox + 2 * (q & 1023);
                    const bool ok = q >= 0 && (unsigned)iz < (unsigned)p.id[0] && (unsigned)iy < (unsigned)p.id[1] && (unsigned)ix < (unsigned)p.id[2];
                    const int gp = ((img * p.id[0] + iz) * p.id[1] + iy) * p.id[2] + ix;
                    dma16g(ok ? base + (size_t)gp * spb : zero, dst + u * 4096);
                }
            return;
        }
#pragma unroll
        for (int u = 0; u < DMA_MAXUP; ++u)
            if (u < nup) dma16g(d_gp[u] >= 0 ? base + (size_t)d_gp[u] * spb : zero, dst + u * 4096);
    };
    auto issue_weights = [&](int kc, int tg, int ntg, int buf) {     // rows of taps >= ntg (padding of an odd or short group) are zero rows
        const char* wb = p.w + ((size_t)tg * p.N * p.K + kc) * ESZ;
        char* dst = smW + buf * wbytes + wave * 1024;
#pragma unroll
        for (int u = 0; u < DMA_MAXUW; ++u)
            if (u < nuw) dma16g((int)(w_d[u] >> 27) < ntg ? wb + (w_d[u] & 0x7ffffffu) : zero, dst + u * 4096);
    };

    // ---- per-lane fragment read constants ----------------------------------------------------------------------------
    const int mx = (1 << p.tlog[2]) - 1, my = (1 << p.tlog[1]) - 1;
    int a_pix[4];                                // patch pixel of the lane's four tile pixels (tap offset 0)
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        const int pidx = wave * 64 + mf * 16 + lr;
        const int tx = pidx & mx, ty = (pidx >> p.tlog[2]) & my, tz = pidx >> (p.tlog[2] + p.tlog[1]);
        a_pix[mf] = comp ? (tz * pdy + ty) * pdx + tx : ((tz * p.sa) * pdy + ty * p.sa) * pdx + tx * p.sa;
    }
    const int w_lane = CPS == 4 ? lr * 64 + ((g ^ swz_row(lr)) << 4) : lr * 32 + ((g & 1) << 4);

    v4f acc[4][NFRAG];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
        for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = v4f{0.f, 0.f, 0.f, 0.f};

    int kc_begin = 0, kc_end = p.K;
    if (p.ksplit > 1) {
        const int nkc = (p.K + KCE - 1) / KCE, per = (nkc + p.ksplit - 1) / p.ksplit;
        kc_begin = ksp * per * KCE;
        kc_end = min(p.K, kc_begin + per * KCE);
    }

    // ---- stages: (chunk, tap group); the loads of stage s + 1 are issued at the top of stage s ---------------------
    // tap groups of a chunk: uniform groups of tapg slots, or the eight components (their patches differ too)
    const int* const tap_tab = comp ? p.ctap_off : p.tap_off;
    const int ngrp = comp ? 8 : (t_end - t_begin + tapg - 1) / tapg;
    auto grp_first = [&](int gi) { return comp ? p.comp_start[gi] : t_begin + gi * tapg; };
    auto grp_count = [&](int gi) { return comp ? p.comp_start[gi + 1] - p.comp_start[gi] : min(tapg, t_end - (t_begin + gi * tapg)); };
    if (kc_begin < kc_end) { issue_patch(kc_begin, 0, 0); issue_weights(kc_begin, grp_first(0), grp_count(0), 0); }
    int pbuf = 0, wbuf = 0;
    for (int kc = kc_begin; kc < kc_end; kc += KCE) {
        for (int gi = 0; gi < ngrp; ++gi) {
            const int tg = grp_first(gi), ntg = grp_count(gi);
            // this wave's loads of this stage have landed; after the barrier everybody's have, and everybody has left
            // the previous stage, whose buffers the next stage's loads overwrite
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (gi + 1 < ngrp) {
                if (comp) issue_patch(kc, gi + 1, pbuf ^ 1);
                issue_weights(kc, grp_first(gi + 1), grp_count(gi + 1), wbuf ^ 1);
            } else if (kc + KCE < kc_end) { issue_patch(kc + KCE, 0, pbuf ^ 1); issue_weights(kc + KCE, grp_first(0), grp_count(0), wbuf ^ 1); }
            const char* pbase = smP + pbuf * pbytes;
            const char* wbase = smW + wbuf * wbytes + w_lane;
            // the tap offsets of the stage, fetched once; a full group of nine taps takes the unrolled path, where the
            // fragment reads of a step are not tied to the step before it
            int toffs[9];
#pragma unroll
            for (int t = 0; t < 9; ++t) toffs[t] = tap_tab[tg + min(t, ntg - 1)];
            if (CPS == 4) {
                if (ntg == 9) {
#pragma unroll
                    for (int tl = 0; tl < 9; ++tl) {
                        uint4 bfr[4], afr[NFRAG];
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf) {
                            const int ap = a_pix[mf] + toffs[tl];
                            bfr[mf] = *reinterpret_cast<const uint4*>(pbase + ap * 64 + ((g ^ swz_row(ap)) << 4));
                        }
#pragma unroll
                        for (int nf = 0; nf < NFRAG; ++nf)
                            afr[nf] = *reinterpret_cast<const uint4*>(wbase + (tl * NT + nf * 16) * 64);
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                            for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = mma16<TR>(afr[nf], bfr[mf], acc[mf][nf]);
                    }
                } else {
                    for (int tl = 0; tl < ntg; ++tl) {
                        const int toff = tap_tab[tg + tl];
                        uint4 bfr[4], afr[NFRAG];
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf) {
                            const int ap = a_pix[mf] + toff;
                            bfr[mf] = *reinterpret_cast<const uint4*>(pbase + ap * 64 + ((g ^ swz_row(ap)) << 4));
                        }
#pragma unroll
                        for (int nf = 0; nf < NFRAG; ++nf)
                            afr[nf] = *reinterpret_cast<const uint4*>(wbase + (tl * NT + nf * 16) * 64);
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                            for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = mma16<TR>(afr[nf], bfr[mf], acc[mf][nf]);
                    }
                }
            } else {
                // 32-byte rows: the two halves of an MFMA contraction step carry two taps
                if (ntg == 9) {
#pragma unroll
                    for (int s = 0; s < 5; ++s) {
                        const int toff = (g >> 1) ? toffs[s < 4 ? 2 * s + 1 : 8] : toffs[2 * s];     // padding tap: any staged pixel, its weights are zero
                        const int tl = 2 * s + (g >> 1);
                        uint4 bfr[4], afr[NFRAG];
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf)
                            bfr[mf] = *reinterpret_cast<const uint4*>(pbase + (a_pix[mf] + toff) * 32 + ((g & 1) << 4));
#pragma unroll
                        for (int nf = 0; nf < NFRAG; ++nf)
                            afr[nf] = *reinterpret_cast<const uint4*>(wbase + (tl * NT + nf * 16) * 32);
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                            for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = mma16<TR>(afr[nf], bfr[mf], acc[mf][nf]);
                    }
                } else {
                    const int nsteps = (ntg + 1) >> 1;
                    for (int s = 0; s < nsteps; ++s) {
                        const int o0 = tap_tab[tg + 2 * s];
                        const int o1 = tap_tab[tg + min(2 * s + 1, ntg - 1)];
                        const int toff = (g >> 1) ? o1 : o0;
                        const int tl = 2 * s + (g >> 1);
                        uint4 bfr[4], afr[NFRAG];
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf)
                            bfr[mf] = *reinterpret_cast<const uint4*>(pbase + (a_pix[mf] + toff) * 32 + ((g & 1) << 4));
#pragma unroll
                        for (int nf = 0; nf < NFRAG; ++nf)
                            afr[nf] = *reinterpret_cast<const uint4*>(wbase + (tl * NT + nf * 16) * 32);
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                            for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = mma16<TR>(afr[nf], bfr[mf], acc[mf][nf]);
                    }
                }
            }
            wbuf ^= 1;
            if (comp) pbuf ^= 1;
        }
        if (!comp) pbuf ^= 1;
    }
    // every load has been waited for at the top of the last stage; the epilogue's own barrier precedes its use of smem
    igemm_epilogue<TR, NFRAG>(p, acc, smem, img, slot, n0, ph, ksp, bz0, by0, bx0);
}

struct DmaShape { int cps, nfrag, nip, niw; size_t lds; };

unsigned magic_of_d(int d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)d) + 1u; }

// Every shape condition the kernel relies on is checked here and nowhere else.
bool dma_shape(int dtype, const IgemmParams& p, DmaShape* out) {
    static const bool off = getenv("MI_NO_IGEMM_DMA") != nullptr;
    // the stride-2 scatter phases (one to eight taps per phase, the same patch staged by every phase) stay with the
    // register-staged kernel: measured faster there (96^3 network, 64 -> 16 transposed: 28.6 against 35.6 us)
    const bool comp = p.comp_mode != 0 && p.sa == 2;     // stride-2 gather staged by polyphase components
    if (off || (p.sa != 1 && !comp) || p.nphase != 1) return false;
    static const bool comp_off = getenv("MI_NO_IGEMM_DMA_S2") != nullptr;
    if (comp && comp_off) return false;
    const int esz = dtype ? 2 : 4;
    const int kce4 = 64 / esz, kce2 = 32 / esz;
    int cps;
    if (p.K % kce4 == 0 && (p.in0_c >= p.K || p.in0_c % kce4 == 0)) cps = 4;
    else if (p.K % kce2 == 0 && (p.in0_c >= p.K || p.in0_c % kce2 == 0)) cps = 2;
    else return false;
    if (p.N % 16 || p.tlog[0] + p.tlog[1] + p.tlog[2] != 8 || p.tapg < 1 || p.tapg > 9) return false;
    if (comp && (p.cpd[0] > 1023 || p.cpd[1] > 1023 || p.cpd[2] > 1023 || p.comp_start[8] != 27)) return false;
    if (p.out1 && (p.out0_c % 16 || p.out0_c <= 0 || p.out0_c >= p.N)) return false;
    if (p.ksplit > 1 && (!p.kpart || p.N % 4 || p.N / 4 > 256)) return false;
    int gcps, nf;
    igemm_config(dtype, p, &gcps, &nf);          // the channel tile follows the policy of the register-staged kernel
    const int rpi = 64 / cps;
    const long long PP = comp ? (long long)p.cpd[0] * p.cpd[1] * p.cpd[2] : (long long)p.pd[0] * p.pd[1] * p.pd[2];
    const int nip = (int)((PP + rpi - 1) / rpi);
    if (nip > 4 * DMA_MAXUP) return false;
    const int tapg = comp ? 8 : p.tapg;
    const int tapg_pad = cps == 2 ? ((tapg + 1) & ~1) : tapg;
    for (;; nf >>= 1) {
        const int niw = tapg_pad * nf * 16 / rpi;
        const int kce = cps * 16 / esz, nkc = (p.K + kce - 1) / kce;
        const int nkc_wg = p.ksplit > 1 ? (nkc + p.ksplit - 1) / p.ksplit : nkc;
        size_t lds = (size_t)((nkc_wg > 1 || comp) ? 2 : 1) * nip * 1024 + (size_t)2 * niw * 1024;
        if (lds < 4096) lds = 4096;                                              // the epilogue's statistics partials
        if (niw <= 4 * DMA_MAXUW && lds <= 160 * 1024) {
            out->cps = cps; out->nfrag = nf; out->nip = nip; out->niw = niw; out->lds = lds;
            break;
        }
        if (nf == 1) return false;
    }
    // 32-bit pixel indices and weight offsets below the tap field of the descriptor word
    if ((long long)p.nb * p.id[0] * p.id[1] * p.id[2] >= (1ll << 31) || (long long)p.nb * p.od[0] * p.od[1] * p.od[2] >= (1ll << 31)) return false;
    if ((long long)(tapg_pad + 1) * p.N * p.K * esz + 64 >= (1ll << 27)) return false;
    if (PP >= (1 << 20)) return false;
    return true;
}

template <class TR, int CPS, int NFRAG>
int launch_dma_t(const IgemmParams& pin, const DmaShape& sh, hipStream_t st) {
    IgemmParams p = pin;
    if (!p.out1) p.out0_c = p.N;
    if (p.comp_mode && p.sa == 2) { p.mg_pd2 = magic_of_d(p.cpd[2]); p.mg_pd12 = magic_of_d(p.cpd[1] * p.cpd[2]); }
    else { p.mg_pd2 = magic_of_d(p.pd[2]); p.mg_pd12 = magic_of_d(p.pd[1] * p.pd[2]); }
    auto kern = igemm_dma_kernel<TR, CPS, NFRAG>;
    if (sh.lds > 48 * 1024) {
        hipError_t e = mi_func_set_lds(reinterpret_cast<const void*>(kern), (int)sh.lds);
        if (e != hipSuccess) { mi_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return -2; }
    }
    dim3 grid(p.nb * p.ntile[0] * p.ntile[1] * p.ntile[2], p.N / (NFRAG * 16), p.nphase * (p.ksplit > 1 ? p.ksplit : 1));
    hipLaunchKernelGGL(kern, grid, dim3(256), sh.lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("igemm_dma launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

}  // namespace

bool igemm_dma_applicable(int dtype, const IgemmParams& p) {
    DmaShape sh;
    return dma_shape(dtype, p, &sh);
}

void igemm_dma_label(int dtype, const IgemmParams& p, char* buf, size_t n) {
    DmaShape sh;
    if (dma_shape(dtype, p, &sh)) snprintf(buf, n, "igemm_dma_kernel<%s,%d,%d>", dtype ? "BF16" : "F32", sh.cps, sh.nfrag);
}

// channel tile (in 16-channel fragments) and contraction chunk (in 16-byte pieces) the kernel would use for p
bool igemm_dma_config(int dtype, const IgemmParams& p, int* cps, int* nfrag) {
    DmaShape sh;
    if (!dma_shape(dtype, p, &sh)) return false;
    *cps = sh.cps; *nfrag = sh.nfrag;
    return true;
}

int launch_igemm_dma(int dtype, const IgemmParams& p, hipStream_t st) {
    DmaShape sh;
    if (!dma_shape(dtype, p, &sh)) { mi_set_error("launch_igemm_dma: shape not served"); return -1; }
#define DMA_CASE(TRT, C, NF) if (sh.cps == C && sh.nfrag == NF) return launch_dma_t<TRT, C, NF>(p, sh, st);
    if (dtype) {
        DMA_CASE(BF16, 4, 4) DMA_CASE(BF16, 4, 2) DMA_CASE(BF16, 4, 1)
        DMA_CASE(BF16, 2, 4) DMA_CASE(BF16, 2, 2) DMA_CASE(BF16, 2, 1)
    } else {
        DMA_CASE(F32, 4, 4) DMA_CASE(F32, 4, 2) DMA_CASE(F32, 4, 1)
        DMA_CASE(F32, 2, 4) DMA_CASE(F32, 2, 2) DMA_CASE(F32, 2, 1)
    }
#undef DMA_CASE
    mi_set_error("launch_igemm_dma: no kernel variant");
    return -1;
}
