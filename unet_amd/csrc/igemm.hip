// Gather-GEMM convolution kernels for gfx950 (MI355X): forward / data-gradient and
// weight-gradient of 3x3(x3) and 1x1 convolutions and stride-2 transposed convolutions,
// on channels-last tensors, with MFMA (bf16 16x16x32 or exact-fp32 16x16x4).
//
// Forward / dgrad tile: 256 base pixels x (16*NFRAG) output channels per 256-thread
// workgroup.  The input patch (tile + halo) of one contraction chunk is staged in LDS
// once and re-used by every tap; the weight tile of the chunk is staged next to it.
// MFMA roles are swapped on purpose (A = weights, B = pixels) so that every lane ends
// up with consecutive channels of ONE pixel: channels-last stores are 16-byte vectors
// and the instance-norm statistics reduce over the 16 lanes that share a channel.
#include "common.h"
#include "kparams.h"
#include "epilogue.h"

namespace {

template <class TR, int CPS, int NFRAG>
__global__ __launch_bounds__(256) void igemm_kernel(const IgemmParams p) {
    signal_start(p.start_flag, p.start_val);
    constexpr int ESZ = TR::ESZ;
    constexpr int KCB = CPS * 16;    // contraction bytes per pixel per stage
    constexpr int KCE = KCB / ESZ;   // ... in elements
    constexpr int PITCH = KCB + 16;  // LDS row pitch in bytes (pixel rows and weight rows)
    constexpr int NT = NFRAG * 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
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
    const int PP = p.pd[0] * p.pd[1] * p.pd[2];
    char* smA = smem;
    char* smW = smem + PP * PITCH;

    const int mx = (1 << p.tlog[2]) - 1, my = (1 << p.tlog[1]) - 1;
    int a_base[4];
#pragma unroll
    for (int mf = 0; mf < 4; ++mf) {
        const int pidx = wave * 64 + mf * 16 + lr;
        const int tx = pidx & mx, ty = (pidx >> p.tlog[2]) & my, tz = pidx >> (p.tlog[2] + p.tlog[1]);
        a_base[mf] = (((tz * p.sa) * p.pd[1] + ty * p.sa) * p.pd[2] + tx * p.sa) * PITCH;
    }
    const int t_begin = p.phase_start[ph], t_end = p.phase_start[ph + 1];
    const int iz0 = bz0 * p.sa + p.omin[0], iy0 = by0 * p.sa + p.omin[1], ix0 = bx0 * p.sa + p.omin[2];
    const int pd12 = p.pd[1] * p.pd[2];

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
    for (int kc = kc_begin; kc < kc_end; kc += KCE) {
        const char* src;
        int spitch, scoff;
        if (kc < p.in0_c) { src = p.in0; spitch = p.in0_pitch; scoff = p.in0_coff + kc; }
        else { src = p.in1; spitch = p.in1_pitch; scoff = p.in1_coff + kc - p.in0_c; }
        for (int tg = t_begin; tg < t_end; tg += p.tapg) {
            const int ntg = min(p.tapg, t_end - tg);
            __syncthreads();
            if (tg == t_begin) {
                // stage the input patch of this contraction chunk (zero outside the tensor); loads are issued in
                // batches of SU independent 16-byte vectors per thread before anything is written to LDS
                constexpr int SU = 4;
                const int total = PP * CPS;
                for (int base = tid; base < total; base += 256 * SU) {
                    uint4 v[SU];
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int id = base + u * 256;
                        v[u] = make_uint4(0u, 0u, 0u, 0u);
                        if (id < total) {
                            const int pix = id / CPS, cc = id % CPS;
                            const int pz = fast_div(pix, p.mg_pd12); const int r = pix - pz * pd12;
                            const int py = fast_div(r, p.mg_pd2); const int px = r - py * p.pd[2];
                            const int iz = iz0 + pz, iy = iy0 + py, ix = ix0 + px;
                            if ((unsigned)iz < (unsigned)p.id[0] && (unsigned)iy < (unsigned)p.id[1] &&
                                (unsigned)ix < (unsigned)p.id[2]) {
                                const size_t gp = ((size_t)(img * p.id[0] + iz) * p.id[1] + iy) * p.id[2] + ix;
                                v[u] = *reinterpret_cast<const uint4*>(src + (gp * spitch + scoff) * ESZ + cc * 16);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < SU; ++u) {
                        const int id = base + u * 256;
                        if (id < total) *reinterpret_cast<uint4*>(smA + (id / CPS) * PITCH + (id % CPS) * 16) = v[u];
                    }
                }
            }
            // stage the weight tile: rows (tap, channel slot), this chunk's KCB bytes
            const int ntg_pad = (CPS == 2) ? ((ntg + 1) & ~1) : ntg;
            {
                constexpr int WU = 3;
                const int wtotal = ntg_pad * NT * CPS;
                for (int base = tid; base < wtotal; base += 256 * WU) {
                    uint4 v[WU];
#pragma unroll
                    for (int u = 0; u < WU; ++u) {
                        const int id = base + u * 256;
                        v[u] = make_uint4(0u, 0u, 0u, 0u);
                        if (id < wtotal) {
                            const int cc = id % CPS; const int row = id / CPS;
                            const int r = row % NT; const int tl = row / NT;
                            if (tl < ntg) {
                                const int n = n0 + chan_of_row<NFRAG>(r >> 4, r & 15);
                                v[u] = *reinterpret_cast<const uint4*>(p.w + (((size_t)(tg + tl) * p.N + n) * p.K + kc) * ESZ + cc * 16);
                            }
                        }
                    }
#pragma unroll
                    for (int u = 0; u < WU; ++u) {
                        const int id = base + u * 256;
                        if (id < wtotal) *reinterpret_cast<uint4*>(smW + (id / CPS) * PITCH + (id % CPS) * 16) = v[u];
                    }
                }
            }
            __syncthreads();
            const int nsteps = (CPS == 4) ? ntg : (ntg + 1) / 2;
            for (int s = 0; s < nsteps; ++s) {
                int tl, cc, toff;
                if (CPS == 4) {
                    tl = s; cc = g;
                    toff = p.tap_off[tg + s] * PITCH;
                } else {
                    tl = 2 * s + (g >> 1); cc = g & 1;
                    const int o0 = p.tap_off[tg + 2 * s];
                    const int o1 = p.tap_off[tg + min(2 * s + 1, ntg - 1)];
                    toff = ((g >> 1) ? o1 : o0) * PITCH;
                }
                uint4 bfr[4], afr[NFRAG];
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
                    bfr[mf] = *reinterpret_cast<const uint4*>(smA + a_base[mf] + toff + cc * 16);
#pragma unroll
                for (int nf = 0; nf < NFRAG; ++nf)
                    afr[nf] = *reinterpret_cast<const uint4*>(smW + ((tl * NT + nf * 16 + lr) * PITCH) + cc * 16);
#pragma unroll
                for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                    for (int nf = 0; nf < NFRAG; ++nf) acc[mf][nf] = mma16<TR>(afr[nf], bfr[mf], acc[mf][nf]);
            }
        }
    }

    igemm_epilogue<TR, NFRAG>(p, acc, smem, img, slot, n0, ph, ksp, bz0, by0, bx0);
}

// ---------------------------------------------------------------------------------
// forward / dgrad, two-dimensional unit-stride specialisation
// ---------------------------------------------------------------------------------
// ---- contraction split: finishing pass --------------------------------------------------------------------------
// out[pixel][ch] = sum_split partial[split][pixel][ch] (+ bias) (+ addend), stored in the activation type to one or
// two destinations, plus the per-(image, channel) statistics.  Thread = (pixel row, four channels).
template <class TR>
__global__ __launch_bounds__(256) void ksplit_finish_kernel(const IgemmParams p, int npix_img) {
    signal_start(p.start_flag, p.start_val);
    constexpr int ESZ = TR::ESZ;
    extern __shared__ __attribute__((aligned(16))) float sm[];   // [256][8] statistics partials
    const int img = blockIdx.y;
    const int cv_n = p.N / 4;
    const int rows = 256 / cv_n;
    const int cv = threadIdx.x % cv_n, prow = threadIdx.x / cv_n;
    const int ch = cv * 4;
    const size_t npo = (size_t)p.nb * npix_img;
    const int ppb = (npix_img + gridDim.x - 1) / gridDim.x;
    const int pbeg = blockIdx.x * ppb, pend = min(npix_img, pbeg + ppb);
    float s1[4] = {0.f, 0.f, 0.f, 0.f}, s2[4] = {0.f, 0.f, 0.f, 0.f};
    if (prow < rows)
        for (int pix = pbeg + prow; pix < pend; pix += rows) {
            const size_t op = (size_t)img * npix_img + pix;
            float v[4] = {0.f, 0.f, 0.f, 0.f};
            for (int ks = 0; ks < p.ksplit; ++ks) {
                const float4 t = *reinterpret_cast<const float4*>(p.kpart + ((size_t)ks * npo + op) * p.N + ch);
                v[0] += t.x; v[1] += t.y; v[2] += t.z; v[3] += t.w;
            }
            if (p.stats && ch < p.nstat) {
#pragma unroll
                for (int j = 0; j < 4; ++j) { s1[j] += v[j]; s2[j] += v[j] * v[j]; }
            }
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
        }
    if (p.stats) {
#pragma unroll
        for (int j = 0; j < 4; ++j) { sm[threadIdx.x * 8 + j] = s1[j]; sm[threadIdx.x * 8 + 4 + j] = s2[j]; }
        __syncthreads();
        for (int n = rows; n > 1;) {      // fixed-shape tree over the pixel rows
            const int half = (n + 1) >> 1;
            if (prow + half < n && prow < rows) {
#pragma unroll
                for (int e = 0; e < 8; ++e) sm[threadIdx.x * 8 + e] += sm[(threadIdx.x + half * cv_n) * 8 + e];
            }
            __syncthreads();
            n = half;
        }
        const int slot = blockIdx.x % MI_NSLOT;
        for (int i = threadIdx.x; i < 2 * p.nstat; i += 256) {
            const int c = i >> 1, kk = i & 1;
            atomicAdd(&p.stats[(((size_t)img * MI_NSLOT + slot) * p.stats_c + c) * 2 + kk], (double)sm[(c / 4) * 8 + kk * 4 + (c & 3)]);
        }
    }
}


// Same contract as igemm_kernel for 2-D tensors gathered with unit stride (sa = 1) and a 16 x 16 tile.
// KS = 3: 3x3 taps (patch 18 x 18), KS = 1: one tap (16 x 16), KS = 2: the stride-2 scatter phases whose
// taps sit at offsets {0,1}^2 (17 x 17).  With the geometry known at compile time the per-lane state is
// one LDS base and one output offset; a workgroup walks `tpw` consecutive tiles of one image row, keeps
// its staging descriptors in registers, stages the weight image once when a single staging covers it
// and reduces the statistics partials once.
// The phase-loop variants keep four accumulator sets; left alone the compiler spends 184 / 328 registers on them
// (two / one workgroup per CU), under a cap it fits 144 / 218 without spilling (three / two).
template <class TR, int CPS, int NFRAG, int KS, int SA, bool PHL>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(PHL ? (NFRAG == 1 ? 3 : 2) : 1))) void igemm2d_kernel(const IgemmParams p) {
    signal_start(p.start_flag, p.start_val);
    constexpr int ESZ = TR::ESZ;
    constexpr int KCB = CPS * 16, KCE = KCB / ESZ, PITCH = KCB + 16, NT = NFRAG * 16;
    // weight image: 32-byte rows are stored without padding, the two 16-byte halves of a row swapped in every
    // other group of eight rows (the 16 lanes of a fragment read then hit 16 different bank groups); a third
    // less LDS than padded rows, which is what lets two workgroups with a 33 x 33 patch share a CU
    constexpr int WPITCH = CPS == 2 ? KCB : PITCH;
    constexpr int PD = KS == 3 ? 15 * SA + 3 : KS == 2 ? 17 : 16;
    constexpr int PP = PD * PD, TOTAL = PP * CPS, NSLOT = (TOTAL + 255) / 256;
    // PHL (KS = 2 only): one workgroup computes all four scatter phases from one staged patch
    constexpr int NPH = PHL ? 4 : 1;
    constexpr int MAXTAP = KS == 3 ? 9 : KS == 2 ? (PHL ? 9 : 4) : 1;
    constexpr int PS[5] = {0, 1, 3, 5, 9};   // tap ranges of the four 2-D phases (even/even, x odd, y odd, both odd)
    static_assert(!PHL || (KS == 2 && CPS == 4), "the phase loop needs the stride-2 scatter taps and 64-byte chunks");
    constexpr int TAPPAD = (CPS == 2) ? ((MAXTAP + 1) & ~1) : MAXTAP;
    constexpr int MAXW = (TAPPAD * NT * CPS + 255) / 256;
    constexpr int RSTEP = (256 / CPS) * PITCH;     // LDS bytes between two staging slots of a thread
    extern __shared__ __attribute__((aligned(16))) char smem[];
    char* smA = smem;
    char* smW = smem + PP * PITCH;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    const int ntx = p.ntile[2];
    const int wg_per_row = ntx / p.tpw;
    int b = blockIdx.x;
    const int txw = b % wg_per_row; b /= wg_per_row;
    const int tyi = b % p.ntile[1];
    const int img = b / p.ntile[1];
    const int slot = blockIdx.x % MI_NSLOT;
    const int n0 = blockIdx.y * NT;
    const int ph = PHL ? 0 : blockIdx.z;
    const int H = p.id[1], W = p.id[2];
    const int t_begin = PHL ? 0 : p.phase_start[ph], ntap = PHL ? 9 : p.phase_start[ph + 1] - t_begin;
    const bool w_once = p.K <= KCE;

    // lane constants: tile pixel (ty = 4 wave + mf, tx = lr)
    const int a_base0 = ((wave * 4 * SA) * PD + lr * SA) * PITCH;
    const int rel0 = (wave * 4 * p.so) * p.od[2] + lr * p.so;
    const int rel_step = p.so * p.od[2];
    const int cc_s = tid % CPS;
    const int lo0 = (tid / CPS) * PITCH + cc_s * 16;
    int s_pl[NSLOT], s_co[NSLOT];
#pragma unroll
    for (int u = 0; u < NSLOT; ++u) {
        const int pix = tid / CPS + u * (256 / CPS);
        const int py = pix / PD, px = pix - py * PD;
        s_pl[u] = py * W + px;
        s_co[u] = (u * 256 + tid < TOTAL) ? ((py << 10) | px) : -1;
    }
    unsigned w_d[MAXW];
#pragma unroll
    for (int u = 0; u < MAXW; ++u) {
        const int row = tid / CPS + u * (256 / CPS);
        const int r = row % NT, tl = row / NT;
        const int n = n0 + chan_of_row<NFRAG>(r >> 4, r & 15);
        w_d[u] = ((unsigned)tl << 27) | (unsigned)(((tl * p.N + n) * p.K) * ESZ + cc_s * 16);
    }
    const int ntap_pad = (CPS == 2) ? ((ntap + 1) & ~1) : ntap;
    auto stage_weights = [&](int kc) {
        const char* wb = p.w + ((size_t)t_begin * p.N * p.K + kc) * ESZ;
        uint4 v[MAXW];
#pragma unroll
        for (int u = 0; u < MAXW; ++u) {
            v[u] = make_uint4(0u, 0u, 0u, 0u);
            if ((int)(w_d[u] >> 27) < ntap) v[u] = *reinterpret_cast<const uint4*>(wb + (w_d[u] & 0x7ffffffu));
        }
#pragma unroll
        for (int u = 0; u < MAXW; ++u)
            if ((int)(w_d[u] >> 27) < ntap_pad) {
                const int row = tid / CPS + u * (256 / CPS);
                const int cw = CPS == 2 ? (cc_s ^ ((row >> 3) & 1)) : cc_s;
                *reinterpret_cast<uint4*>(smW + row * WPITCH + cw * 16) = v[u];
            }
    };
    if (w_once) stage_weights(0);
    int toffs[MAXTAP];
#pragma unroll
    for (int t = 0; t < MAXTAP; ++t) toffs[t] = p.tap_off[t_begin + (t < ntap ? t : 0)] * PITCH;

    const bool do_stats = p.stats != nullptr;
    float s1[NFRAG][4], s2[NFRAG][4];
#pragma unroll
    for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
        for (int j = 0; j < 4; ++j) { s1[nf][j] = 0.f; s2[nf][j] = 0.f; }

    const int by0 = tyi << 4;
    const int iy0 = by0 * SA + p.omin[1];
    for (int ti = 0; ti < p.tpw; ++ti) {
        const int bx0 = (txw * p.tpw + ti) << 4;
        const int ix0 = bx0 * SA + p.omin[2];
        const bool interior = iy0 >= 0 && ix0 >= 0 && iy0 + PD <= H && ix0 + PD <= W;
        const long long org = ((long long)img * H + iy0) * W + ix0;   // patch origin pixel (may lie outside)

        v4f acc[NPH][4][NFRAG];
#pragma unroll
        for (int q = 0; q < NPH; ++q)
#pragma unroll
            for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                for (int nf = 0; nf < NFRAG; ++nf) acc[q][mf][nf] = v4f{0.f, 0.f, 0.f, 0.f};

        for (int kc = 0; kc < p.K; kc += KCE) {
            const char* src;
            int spitch, scoff;
            if (kc < p.in0_c) { src = p.in0; spitch = p.in0_pitch; scoff = p.in0_coff + kc; }
            else { src = p.in1; spitch = p.in1_pitch; scoff = p.in1_coff + kc - p.in0_c; }
            const int spb = spitch * ESZ;
            const char* pb = src + org * spb + scoff * ESZ + cc_s * 16;   // only dereferenced inside the tensor
            __syncthreads();
            {
                constexpr int SBATCH = 6;
#pragma unroll
                for (int u0 = 0; u0 < NSLOT; u0 += SBATCH) {
                    uint4 v[SBATCH];
#pragma unroll
                    for (int u = u0; u < u0 + SBATCH && u < NSLOT; ++u) {
                        v[u - u0] = make_uint4(0u, 0u, 0u, 0u);
                        bool ok = s_co[u] >= 0;
                        if (!interior && ok) {
                            const int iy = iy0 + (s_co[u] >> 10), ix = ix0 + (s_co[u] & 1023);
                            ok = (unsigned)iy < (unsigned)H && (unsigned)ix < (unsigned)W;
                        }
                        if (ok) v[u - u0] = *reinterpret_cast<const uint4*>(pb + (size_t)__umul24(s_pl[u], spb));
                    }
#pragma unroll
                    for (int u = u0; u < u0 + SBATCH && u < NSLOT; ++u)
                        if (s_co[u] >= 0) *reinterpret_cast<uint4*>(smA + lo0 + u * RSTEP) = v[u - u0];
                }
            }
            if (!w_once) stage_weights(kc);
            __syncthreads();
            if (PHL) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int t = PS[q]; t < PS[q + 1]; ++t) {
                        uint4 bfr[4], afr[NFRAG];
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf)
                            bfr[mf] = *reinterpret_cast<const uint4*>(smA + a_base0 + mf * SA * PD * PITCH + toffs[t] + g * 16);
#pragma unroll
                        for (int nf = 0; nf < NFRAG; ++nf)
                            afr[nf] = *reinterpret_cast<const uint4*>(smW + ((t * NT + nf * 16 + lr) * WPITCH) + g * 16);
#pragma unroll
                        for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                            for (int nf = 0; nf < NFRAG; ++nf) acc[q][mf][nf] = mma16<TR>(afr[nf], bfr[mf], acc[q][mf][nf]);
                    }
            } else {
                constexpr int NSTEP = (CPS == 4) ? MAXTAP : (MAXTAP + 1) / 2;
#pragma unroll
                for (int s = 0; s < NSTEP; ++s) {
                    int tl, cc, toff;
                    if (CPS == 4) {
                        if (s >= ntap) break;
                        tl = s; cc = g; toff = toffs[s];
                    } else {
                        if (2 * s >= ntap) break;
                        tl = 2 * s + (g >> 1); cc = g & 1;
                        toff = (g >> 1) ? toffs[(2 * s + 1 < MAXTAP) ? 2 * s + 1 : 2 * s] : toffs[2 * s];
                    }
                    uint4 bfr[4], afr[NFRAG];
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf)
                        bfr[mf] = *reinterpret_cast<const uint4*>(smA + a_base0 + mf * SA * PD * PITCH + toff + cc * 16);
#pragma unroll
                    for (int nf = 0; nf < NFRAG; ++nf)
                        afr[nf] = *reinterpret_cast<const uint4*>(smW + ((tl * NT + nf * 16 + lr) * WPITCH) +
                                                                  (CPS == 2 ? (cc ^ ((lr >> 3) & 1)) : cc) * 16);
#pragma unroll
                    for (int mf = 0; mf < 4; ++mf)
#pragma unroll
                        for (int nf = 0; nf < NFRAG; ++nf) acc[0][mf][nf] = mma16<TR>(afr[nf], bfr[mf], acc[0][mf][nf]);
                }
            }
        }

        // ---- tile epilogue ---------------------------------------------------------------------
        const bool full = by0 + 16 <= p.bd[1] && bx0 + 16 <= p.bd[2];
#pragma unroll
        for (int q = 0; q < NPH; ++q) {
        const int phx = PHL ? q : ph;
        const int base_pix = (img * p.od[1] + by0 * p.so + p.phase_off[phx][1]) * p.od[2] + bx0 * p.so + p.phase_off[phx][2];
#pragma unroll
        for (int mf = 0; mf < 4; ++mf) {
            const bool valid = full || (by0 + wave * 4 + mf < p.bd[1] && bx0 + lr < p.bd[2]);
            if (do_stats) {
#pragma unroll
                for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const float v = valid ? acc[q][mf][nf][j] : 0.f;
                        s1[nf][j] += v; s2[nf][j] += v * v;
                    }
            }
            if (valid) {
                const size_t op = (size_t)(base_pix + rel0 + mf * rel_step);
                if (NFRAG == 1) {
                    const int ch = n0 + 4 * g;
                    float v[4];
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = acc[q][mf][0][j];
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
                        for (int j = 0; j < 4; ++j) { v[j] = acc[q][mf][2 * pf][j]; v[4 + j] = acc[q][mf][2 * pf + 1][j]; }
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
        }
    }
    if (do_stats) {
        __syncthreads();
        float* sst = reinterpret_cast<float*>(smem);  // [4 waves][NT][2]
#pragma unroll
        for (int nf = 0; nf < NFRAG; ++nf)
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                const float a = row_sum16(s1[nf][j]), bsum = row_sum16(s2[nf][j]);
                if (lr == 0) {
                    const int cl = chan_of_row<NFRAG>(nf, 4 * g + j);
                    sst[(wave * NT + cl) * 2] = a;
                    sst[(wave * NT + cl) * 2 + 1] = bsum;
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

// ---------------------------------------------------------------------------------
// weight gradient
// ---------------------------------------------------------------------------------
// Workgroup = 4 waves arranged WK x WF x WS: WF x WS waves tile the (F, S) channel block,
// WK waves split the pixels (contraction) of each staged tile.  Every wave owns
// FF x SF 16x16 fragments for each of the NT taps.
template <class TR, int NT, int FF, int SF, int WF, int WS>
__global__ __launch_bounds__(256) void wgrad_kernel(const WgradParams p) {
    constexpr int ESZ = TR::ESZ;
    constexpr int WK = 4 / (WF * WS);
    constexpr int FB = 16 * FF * WF, SB = 16 * SF * WS;
    constexpr int FPITCH = FB * ESZ + 16, SPITCH = SB * ESZ + 16;
    extern __shared__ __attribute__((aligned(16))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int lr = lane & 15, g = lane >> 4;
    const int ws = wave % WS, wf = (wave / WS) % WF, wk = wave / (WS * WF);
    const int fb0 = blockIdx.y * FB;
    const int sblk = blockIdx.z / p.ntapg, tgi = blockIdx.z % p.ntapg;
    const int sb0 = sblk * SB;
    const int PP = p.pd[0] * p.pd[1] * p.pd[2];
    const int pd12 = p.pd[1] * p.pd[2];
    char* smF = smem;
    char* smS = smem + 256 * FPITCH;
    const int mx = (1 << p.tlog[2]) - 1, my = (1 << p.tlog[1]) - 1;

    v4f acc[NT][FF][SF];
    v4f accb[FF];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int a = 0; a < FF; ++a)
#pragma unroll
            for (int b = 0; b < SF; ++b) acc[t][a][b] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < FF; ++a) accb[a] = v4f{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (ws == 0) && (blockIdx.z == 0);

    const char* fsrc; int fpitch, fcoff;
    if (fb0 < p.f0_c) { fsrc = p.f0; fpitch = p.f0_pitch; fcoff = p.f0_coff + fb0; }
    else { fsrc = p.f1; fpitch = p.f1_pitch; fcoff = p.f1_coff + fb0 - p.f0_c; }
    const int tiles_per_img = p.ntile[0] * p.ntile[1] * p.ntile[2];

    // One tile = 256 base pixels of one image.  load_tile() issues the global loads of a tile into registers
    // (F: FVN vectors per thread; S patch: up to SMAX), store_tile() writes them to LDS.  When the patch fits
    // the register budget the loads of tile i+1 are in flight while tile i is multiplied.
    constexpr int FVPP = FB * ESZ / 16;      // 16-byte vectors per F pixel = F vectors per thread
    constexpr int SVPP = SB * ESZ / 16;
    constexpr int SMAX = 12;
    const int stotal = PP * SVPP;
    const int nsv = (stotal + 255) >> 8;
    uint4 fv[FVPP], sv[SMAX];
    auto load_tile = [&](int tile, bool with_s) {
        const int img = tile / tiles_per_img;
        int tr = tile - img * tiles_per_img;
        const int txi = tr % p.ntile[2]; tr /= p.ntile[2];
        const int tyi = tr % p.ntile[1]; const int tzi = tr / p.ntile[1];
        const int bz0 = tzi << p.tlog[0], by0 = tyi << p.tlog[1], bx0 = txi << p.tlog[2];
        const int iz0 = bz0 * p.ss + p.omin[0] + p.tg_zoff[tgi], iy0 = by0 * p.ss + p.omin[1],
                  ix0 = bx0 * p.ss + p.omin[2];
#pragma unroll
        for (int u = 0; u < FVPP; ++u) {
            const int id = tid + u * 256;
            const int pix = id / FVPP, cv = id % FVPP;
            const int tx = pix & mx, ty = (pix >> p.tlog[2]) & my, tz = pix >> (p.tlog[2] + p.tlog[1]);
            const int bz = bz0 + tz, by = by0 + ty, bx = bx0 + tx;
            fv[u] = make_uint4(0u, 0u, 0u, 0u);
            if (bz < p.bd[0] && by < p.bd[1] && bx < p.bd[2]) {
                const size_t gp = ((size_t)(img * p.bd[0] + bz) * p.bd[1] + by) * p.bd[2] + bx;
                fv[u] = *reinterpret_cast<const uint4*>(fsrc + (gp * fpitch + fcoff) * ESZ + cv * 16);
            }
        }
        if (with_s) {
#pragma unroll
            for (int u = 0; u < SMAX; ++u) {
                const int id = tid + u * 256;
                sv[u] = make_uint4(0u, 0u, 0u, 0u);
                if (id < stotal) {
                    const int pix = id / SVPP, cv = id % SVPP;
                    const int pz = fast_div(pix, p.mg_pd12); const int r = pix - pz * pd12;
                    const int py = fast_div(r, p.mg_pd2); const int px = r - py * p.pd[2];
                    const int iz = iz0 + pz, iy = iy0 + py, ix = ix0 + px;
                    if ((unsigned)iz < (unsigned)p.sd[0] && (unsigned)iy < (unsigned)p.sd[1] &&
                        (unsigned)ix < (unsigned)p.sd[2]) {
                        const size_t gp = ((size_t)(img * p.sd[0] + iz) * p.sd[1] + iy) * p.sd[2] + ix;
                        sv[u] = *reinterpret_cast<const uint4*>(p.s + (gp * p.s_pitch + p.s_coff + sb0) * ESZ + cv * 16);
                    }
                }
            }
        }
    };
    auto store_tile = [&](bool with_s) {
#pragma unroll
        for (int u = 0; u < FVPP; ++u) {
            const int id = tid + u * 256;
            *reinterpret_cast<uint4*>(smF + (id / FVPP) * FPITCH + (id % FVPP) * 16) = fv[u];
        }
        if (with_s) {
#pragma unroll
            for (int u = 0; u < SMAX; ++u) {
                const int id = tid + u * 256;
                if (id < stotal) *reinterpret_cast<uint4*>(smS + (id / SVPP) * SPITCH + (id % SVPP) * 16) = sv[u];
            }
        }
    };
    // S patch staged straight to LDS in batches (patches too large for the register budget)
    auto stage_s_direct = [&](int tile) {
        const int img = tile / tiles_per_img;
        int tr = tile - img * tiles_per_img;
        const int txi = tr % p.ntile[2]; tr /= p.ntile[2];
        const int tyi = tr % p.ntile[1]; const int tzi = tr / p.ntile[1];
        const int iz0 = (tzi << p.tlog[0]) * p.ss + p.omin[0] + p.tg_zoff[tgi], iy0 = (tyi << p.tlog[1]) * p.ss + p.omin[1],
                  ix0 = (txi << p.tlog[2]) * p.ss + p.omin[2];
        constexpr int SU = 4;
        for (int base = tid; base < stotal; base += 256 * SU) {
            uint4 v[SU];
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int id = base + u * 256;
                v[u] = make_uint4(0u, 0u, 0u, 0u);
                if (id < stotal) {
                    const int pix = id / SVPP, cv = id % SVPP;
                    const int pz = fast_div(pix, p.mg_pd12); const int r = pix - pz * pd12;
                    const int py = fast_div(r, p.mg_pd2); const int px = r - py * p.pd[2];
                    const int iz = iz0 + pz, iy = iy0 + py, ix = ix0 + px;
                    if ((unsigned)iz < (unsigned)p.sd[0] && (unsigned)iy < (unsigned)p.sd[1] &&
                        (unsigned)ix < (unsigned)p.sd[2]) {
                        const size_t gp = ((size_t)(img * p.sd[0] + iz) * p.sd[1] + iy) * p.sd[2] + ix;
                        v[u] = *reinterpret_cast<const uint4*>(p.s + (gp * p.s_pitch + p.s_coff + sb0) * ESZ + cv * 16);
                    }
                }
            }
#pragma unroll
            for (int u = 0; u < SU; ++u) {
                const int id = base + u * 256;
                if (id < stotal) *reinterpret_cast<uint4*>(smS + (id / SVPP) * SPITCH + (id % SVPP) * 16) = v[u];
            }
        }
    };
    const bool pf_s = nsv <= SMAX;
    if (blockIdx.x < p.tiles_total) load_tile(blockIdx.x, pf_s);
    for (int tile = blockIdx.x; tile < p.tiles_total; tile += p.nsplit) {
        __syncthreads();                       // the previous tile has been consumed
        store_tile(pf_s);
        if (!pf_s) stage_s_direct(tile);
        __syncthreads();
        if (tile + p.nsplit < p.tiles_total) load_tile(tile + p.nsplit, pf_s);   // in flight during the multiply
        for (int ks = wk; ks < 8; ks += WK) {
            if (TR::ESZ == 2) {
                // bf16: K-step = 32 pixels; lane group g covers pixels 8g..8g+7 via two transposed LDS reads
                const int q = lr >> 2, pp = lr & 3;
                int soff[2];
                uint4 afr[FF];
#pragma unroll
                for (int h = 0; h < 2; ++h) {
                    const int pix = ks * 32 + 8 * g + 4 * h + q;
                    const int tx = pix & mx, ty = (pix >> p.tlog[2]) & my, tz = pix >> (p.tlog[2] + p.tlog[1]);
                    soff[h] = (((tz * p.ss) * p.pd[1] + ty * p.ss) * p.pd[2] + tx * p.ss) * SPITCH + pp * 8;
                }
#pragma unroll
                for (int a = 0; a < FF; ++a) {
                    const int c0 = ((wf * FF + a) * 16 + pp * 4) * 2;
                    const int p0 = ks * 32 + 8 * g + q;
                    v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) v4s*)(smF + p0 * FPITCH + c0));
                    v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                        (__attribute__((address_space(3))) v4s*)(smF + (p0 + 4) * FPITCH + c0));
                    v8s f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    afr[a] = __builtin_bit_cast(uint4, f);
                }
                if (do_bias) {
                    const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
#pragma unroll
                    for (int a = 0; a < FF; ++a) accb[a] = mma16<BF16>(afr[a], ones, accb[a]);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int toff = p.tap_off[t] * SPITCH;
                    uint4 bfr[SF];
#pragma unroll
                    for (int b = 0; b < SF; ++b) {
                        const int c0 = ((ws * SF + b) * 16) * 2;
                        v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) v4s*)(smS + soff[0] + toff + c0));
                        v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16(
                            (__attribute__((address_space(3))) v4s*)(smS + soff[1] + toff + c0));
                        v8s f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        bfr[b] = __builtin_bit_cast(uint4, f);
                    }
#pragma unroll
                    for (int a = 0; a < FF; ++a)
#pragma unroll
                        for (int b = 0; b < SF; ++b) acc[t][a][b] = mma16<BF16>(afr[a], bfr[b], acc[t][a][b]);
                }
            } else {
                // fp32: eight 16x16x4 MFMAs per 32-pixel step; lane (row lr, k-slot g) reads pixel 4i+g
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int pix = ks * 32 + 4 * i + g;
                    const int tx = pix & mx, ty = (pix >> p.tlog[2]) & my, tz = pix >> (p.tlog[2] + p.tlog[1]);
                    const int soff = (((tz * p.ss) * p.pd[1] + ty * p.ss) * p.pd[2] + tx * p.ss) * SPITCH;
                    float av[FF];
#pragma unroll
                    for (int a = 0; a < FF; ++a)
                        av[a] = *reinterpret_cast<const float*>(smF + pix * FPITCH + ((wf * FF + a) * 16 + lr) * 4);
                    if (do_bias) {
#pragma unroll
                        for (int a = 0; a < FF; ++a)
                            accb[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], 1.0f, accb[a], 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int toff = p.tap_off[t] * SPITCH;
#pragma unroll
                        for (int b = 0; b < SF; ++b) {
                            const float bv =
                                *reinterpret_cast<const float*>(smS + soff + toff + ((ws * SF + b) * 16 + lr) * 4);
#pragma unroll
                            for (int a = 0; a < FF; ++a)
                                acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv, acc[t][a][b], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- combine the WK pixel-split waves in a fixed order through LDS, then write one slab -------
    if (WK > 1) {
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);   // [WK][(NT*FF*SF + FF)*4][64 lanes]
        constexpr int NV = (NT * FF * SF + FF) * 4;
        const int sub = wave % (WF * WS);              // waves that share (wf, ws) hold the same output fragment set
        float* mine = red + ((size_t)(wk * WF * WS + sub) * NV) * 64 + lane;
        int v = 0;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int a = 0; a < FF; ++a)
#pragma unroll
                for (int b = 0; b < SF; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mine[(v++) * 64] = acc[t][a][b][j];
#pragma unroll
        for (int a = 0; a < FF; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) mine[(v++) * 64] = accb[a][j];
        __syncthreads();
        if (wk == 0) {
            v = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int a = 0; a < FF; ++a)
#pragma unroll
                    for (int b = 0; b < SF; ++b)
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            float r = acc[t][a][b][j];
                            for (int k = 1; k < WK; ++k) r += red[((size_t)(k * WF * WS + sub) * NV + v) * 64 + lane];
                            acc[t][a][b][j] = r; ++v;
                        }
#pragma unroll
            for (int a = 0; a < FF; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    float r = accb[a][j];
                    for (int k = 1; k < WK; ++k) r += red[((size_t)(k * WF * WS + sub) * NV + v) * 64 + lane];
                    accb[a][j] = r; ++v;
                }
        }
    }
    if (wk == 0) {
        const size_t slab = (size_t)blockIdx.x;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const size_t tbase = (slab * p.ntaps_total + (size_t)tgi * NT + t) * p.F;
#pragma unroll
            for (int a = 0; a < FF; ++a)
#pragma unroll
                for (int b = 0; b < SF; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = fb0 + (wf * FF + a) * 16 + 4 * g + j;
                        const int s = sb0 + (ws * SF + b) * 16 + lr;
                        p.partial[(tbase + f) * p.S + s] = acc[t][a][b][j];
                    }
        }
        if (do_bias && lr == 0) {
#pragma unroll
            for (int a = 0; a < FF; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j)
                    p.partial_bias[slab * p.F + fb0 + (wf * FF + a) * 16 + 4 * g + j] = accb[a][j];
        }
    }
}

// ---------------------------------------------------------------------------------
// slab reduction and weight packing (descriptor tables in device memory)
// ---------------------------------------------------------------------------------
__device__ __forceinline__ void reduce_body(const ReduceDesc& d, int blk, int nblk);
template <class TR>
__device__ __forceinline__ void pack_body(const PackDesc& d, int blk, int nblk);
// one launch for a table of descriptors: workgroup b serves the descriptor whose range [blk0, blk0 + nblk) holds b
__global__ __launch_bounds__(256) void reduce_kernel(const ReduceTable tab) {
    int i = 0;
    while (i + 1 < tab.n && (int)blockIdx.x >= tab.d[i + 1].blk0) ++i;
    reduce_body(tab.d[i], (int)blockIdx.x - tab.d[i].blk0, tab.d[i].nblk);
}
template <class TR>
__global__ __launch_bounds__(256) void pack_kernel(const PackTable tab) {
    int i = 0;
    while (i + 1 < tab.n && (int)blockIdx.x >= tab.d[i + 1].blk0) ++i;
    pack_body<TR>(tab.d[i], (int)blockIdx.x - tab.d[i].blk0, tab.d[i].nblk);
}

// 256 threads = EL output elements x SG slab groups (EL = 64, 16, 4 or 1: few outputs with many slabs get more
// slab groups).  Each thread sums its slabs in slab order with eight loads in flight, the group sums are
// combined by a fixed-shape tree through LDS, so the result is bitwise reproducible.
__device__ __forceinline__ void reduce_body(const ReduceDesc& d, int blk, int nblk) {
    __shared__ float part[256];
    const int nw = d.ntaps * d.Fcnt * d.S;
    const int total = nw + (d.dst_b ? d.Fcnt : 0);
    const int EL = total >= 2048 ? 64 : total >= 256 ? 16 : total >= 16 ? 4 : 1;
    const int SG = 256 / EL;
    const int lx = threadIdx.x % EL, ly = threadIdx.x / EL;
    const int nslab = d.nslab;
    const size_t stride = (size_t)d.ntaps * d.Ftot * d.S;
    // Few slabs, all taps of the kernel: tiles of P consecutive (f, s) pairs through LDS (P x taps <= 1024).  A tile's
    // slab elements are contiguous per tap (read with consecutive lanes), its results are one contiguous piece of the
    // gradient tensor (pairs x taps floats): written tap by tap straight from the sums they would be four-byte
    // stores at a pitch of `taps` floats.  A thread sums up to four elements, each in slab order (reproducible), with
    // the loads of all four and of four slabs in flight together.
    const int pairs = d.Fcnt * d.S;
    int P = d.KK <= 4 ? 256 : d.KK <= 16 ? 64 : 32;
    while (P > 16 && pairs % P) P >>= 1;
    if (d.ntaps == d.KK && d.KK <= 27 && nslab <= d.tiled_max_slabs && pairs % P == 0) {
        __shared__ float tile[1024];
        const int KK = d.KK, ntile = pairs / P, ne = P * KK;
        for (int tix = blk; tix < ntile; tix += nblk) {
            const int p0 = tix * P;
            const float* src[4]; int dsti[4]; bool ok[4]; float a[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                const int e = threadIdx.x + 256 * q;
                ok[q] = e < ne;
                const int t = ok[q] ? e / P : 0, pl = ok[q] ? e - t * P : 0;
                src[q] = d.partial + ((size_t)t * d.Ftot + d.f_off) * d.S + p0 + pl;
                dsti[q] = pl * KK + d.kidx[t];
                a[q] = 0.f;
            }
            int sl = 0;
            for (; sl + 3 < nslab; sl += 4) {
                float v[4][4];
#pragma unroll
                for (int q = 0; q < 4; ++q)
#pragma unroll
                    for (int k = 0; k < 4; ++k) v[q][k] = ok[q] ? src[q][(size_t)(sl + k) * stride] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] = (((a[q] + v[q][0]) + v[q][1]) + v[q][2]) + v[q][3];
            }
            for (; sl < nslab; ++sl) {
                float v[4];
#pragma unroll
                for (int q = 0; q < 4; ++q) v[q] = ok[q] ? src[q][(size_t)sl * stride] : 0.f;
#pragma unroll
                for (int q = 0; q < 4; ++q) a[q] += v[q];
            }
            __syncthreads();                        // the previous tile has been written out
#pragma unroll
            for (int q = 0; q < 4; ++q) if (ok[q]) tile[dsti[q]] = a[q];
            __syncthreads();
            float* dst = d.dst_w + (size_t)p0 * KK;
            for (int e = threadIdx.x; e < ne; e += 256) dst[e] = tile[e];
        }
        if (d.dst_b)
            for (int f = blk * 256 + threadIdx.x; f < d.Fcnt; f += nblk * 256) {
                const float* sb = d.partial_bias + d.f_off + f;
                float a = 0.f;
                for (int sl = 0; sl < nslab; ++sl) a += sb[(size_t)sl * d.Ftot];
                d.dst_b[f] = a;
            }
        return;
    }
    for (int base = blk * EL; base < total; base += nblk * EL) {
        const int i = base + lx;
        float a = 0.f;
        int s = 0, f = 0, t = 0;
        const float* src = nullptr;
        size_t st = 0;
        if (i < nw) {
            s = i % d.S; const int r = i / d.S;
            f = r % d.Fcnt; t = r / d.Fcnt;
            src = d.partial + ((size_t)t * d.Ftot + d.f_off + f) * d.S + s;
            st = stride;
        } else if (i < total) {
            f = i - nw;
            src = d.partial_bias + d.f_off + f;
            st = (size_t)d.Ftot;
        }
        if (src) {
            int sl = ly;
            for (; sl + 7 * SG < nslab; sl += 8 * SG) {
                const float v0 = src[(size_t)sl * st], v1 = src[(size_t)(sl + SG) * st];
                const float v2 = src[(size_t)(sl + 2 * SG) * st], v3 = src[(size_t)(sl + 3 * SG) * st];
                const float v4 = src[(size_t)(sl + 4 * SG) * st], v5 = src[(size_t)(sl + 5 * SG) * st];
                const float v6 = src[(size_t)(sl + 6 * SG) * st], v7 = src[(size_t)(sl + 7 * SG) * st];
                a = (((((((a + v0) + v1) + v2) + v3) + v4) + v5) + v6) + v7;
            }
            for (; sl < nslab; sl += SG) a += src[(size_t)sl * st];
        }
        part[ly * EL + lx] = a;
        __syncthreads();
        for (int step = SG >> 1; step >= 1; step >>= 1) {
            if (ly < step) part[ly * EL + lx] += part[(ly + step) * EL + lx];
            __syncthreads();
        }
        if (ly == 0 && i < total) {
            const float v = part[lx];
            if (i < nw) d.dst_w[((size_t)f * d.S + s) * d.KK + d.kidx[t]] = v;
            else d.dst_b[f] = v;
        }
        __syncthreads();
    }
}
__global__ __launch_bounds__(256) void reduce_one_kernel(const ReduceDesc d) { reduce_body(d, blockIdx.x, gridDim.x); }

// A thread takes one (n, k) pair and walks its taps: the taps of a pair are neighbours in the source (one or two
// cache lines, fetched once), and for every tap the lanes of a wave write consecutive k.
template <class TR>
__device__ __forceinline__ void pack_body(const PackDesc& d, int blk, int nblk) {
    const int pairs = d.N * d.K;
    typename TR::elem* dst = reinterpret_cast<typename TR::elem*>(d.dst);
    // Tiles of 16 x 16 (n, k) pairs through LDS when the source is a dense [n][k][taps] or [k][n][taps] tensor: the
    // 16 pairs of a run along the source's inner dimension are one contiguous piece of 16 * taps floats, read with
    // consecutive lanes; each thread then takes the taps of its own pair from LDS (run pitch and tap count odd: no
    // bank conflicts) and stores them with k fastest.  One pair per thread straight from global memory reads four
    // bytes per lane at a stride of `taps` floats, or of a whole source row, in every load.
    const int KK = d.sn < d.sk ? d.sn : d.sk;
    const bool inner_k = d.sk == KK && d.sn == d.K * KK, inner_n = !inner_k && d.sn == KK && d.sk == d.N * KK;
    if ((inner_k || inner_n) && d.N % 16 == 0 && d.K % 16 == 0 && KK >= 1 && KK <= 27) {
        __shared__ float tile[16 * (16 * 27 + 1)];
        const int run_len = 16 * KK, RP = run_len + 1;
        const int tk = d.K / 16, ntile = (d.N / 16) * tk;
        const int kl = threadIdx.x & 15, nl = threadIdx.x >> 4;
        const float* pv = tile + (inner_k ? nl * RP + kl * KK : kl * RP + nl * KK);
        const size_t ot = (size_t)d.Ntot * d.Ktot;
        for (int tix = blk; tix < ntile; tix += nblk) {
            const int n0 = (tix / tk) * 16, k0 = (tix % tk) * 16;
            __syncthreads();
            for (int e = threadIdx.x; e < 16 * run_len; e += 256) {
                const int run = e / run_len, r = e - run * run_len;
                const float* src = inner_k ? d.src + (size_t)(n0 + run) * d.sn + (size_t)k0 * KK + r
                                           : d.src + (size_t)(k0 + run) * d.sk + (size_t)n0 * KK + r;
                tile[run * RP + r] = *src;
            }
            __syncthreads();
            const size_t o0 = (size_t)(d.n_off + n0 + nl) * d.Ktot + d.k_off + k0 + kl;
            for (int t = 0; t < d.ntaps; ++t) {
                const float v = pv[d.kidx[t]];
                if (TR::ESZ == 2) dst[o0 + (size_t)t * ot] = f32_to_bf16(v);
                else reinterpret_cast<float*>(dst)[o0 + (size_t)t * ot] = v;
            }
        }
        return;
    }
    for (int i = blk * 256 + threadIdx.x; i < pairs; i += nblk * 256) {
        const int n = i / d.K; const int k = i - n * d.K;
        const float* src = d.src + (size_t)n * d.sn + (size_t)k * d.sk;
        const size_t o0 = (size_t)(d.n_off + n) * d.Ktot + d.k_off + k;
        const size_t ot = (size_t)d.Ntot * d.Ktot;
        for (int t0 = 0; t0 < d.ntaps; t0 += 9) {
            float v[9];
#pragma unroll
            for (int j = 0; j < 9; ++j) v[j] = t0 + j < d.ntaps ? src[d.kidx[t0 + j]] : 0.f;
#pragma unroll
            for (int j = 0; j < 9; ++j) {
                if (t0 + j >= d.ntaps) break;
                const size_t o = o0 + (size_t)(t0 + j) * ot;
                if (TR::ESZ == 2) dst[o] = f32_to_bf16(v[j]);
                else reinterpret_cast<float*>(dst)[o] = v[j];
            }
        }
    }
}
template <class TR>
__global__ __launch_bounds__(256) void pack_one_kernel(const PackDesc d) { pack_body<TR>(d, blockIdx.x, gridDim.x); }

template <class K>
int set_lds(K kern, size_t bytes) {
    if (bytes > 160 * 1024) { mi_set_error("LDS request exceeds 160 KiB"); return -1; }
    if (bytes > 48 * 1024) {
        hipError_t e = mi_func_set_lds(reinterpret_cast<const void*>(kern), (int)bytes);
        if (e != hipSuccess) { mi_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return -2; }
    }
    return 0;
}

unsigned magic_of(int d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)d) + 1u; }

template <class TR, int CPS, int NFRAG>
int launch_igemm_t(const IgemmParams& pin, hipStream_t st) {
    IgemmParams p = pin;
    if (!p.out1) p.out0_c = p.N;
    p.mg_pd2 = magic_of(p.pd[2]); p.mg_pd12 = magic_of(p.pd[1] * p.pd[2]);
    constexpr int PITCH = CPS * 16 + 16;
    const int PP = p.pd[0] * p.pd[1] * p.pd[2];
    const int tapg_pad = (CPS == 2) ? ((p.tapg + 1) & ~1) : p.tapg;
    size_t lds = (size_t)PP * PITCH + (size_t)tapg_pad * NFRAG * 16 * PITCH;
    if (lds < (size_t)NFRAG * 16 * 2 * 4 * 4) lds = NFRAG * 16 * 2 * 4 * 4;
    auto kern = igemm_kernel<TR, CPS, NFRAG>;
    if (int r = set_lds(kern, lds)) return r;
    dim3 grid(p.nb * p.ntile[0] * p.ntile[1] * p.ntile[2], p.N / (NFRAG * 16), p.nphase * (p.ksplit > 1 ? p.ksplit : 1));
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("igemm launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

// 2-D unit-stride specialisation: usable when the generic geometry chose a 16 x 16 tile on a 2-D tensor
inline int igemm2d_ks(const IgemmParams& p) {
    if (p.id[0] != 1 || p.tlog[0] != 0 || p.tlog[1] != 4 || p.tlog[2] != 4 || p.pd[0] != 1) return 0;
    const int ntaps = p.phase_start[p.nphase];
    if (p.sa == 2) return (p.nphase == 1 && p.so == 1 && ntaps == 9 && p.pd[1] == 33 && p.pd[2] == 33) ? 3 : 0;
    if (p.sa != 1) return 0;
    if (p.nphase == 4 && p.so == 2 && p.pd[1] == 17 && p.pd[2] == 17 && ntaps == 9) return 2;
    if (p.nphase == 1 && p.so == 1 && ntaps == 9 && p.pd[1] == 18 && p.pd[2] == 18) return 3;
    if (p.nphase == 1 && p.so == 1 && ntaps == 1 && p.pd[1] == 16 && p.pd[2] == 16) return 1;
    return 0;
}

template <class TR, int CPS, int NFRAG, int KS, int SA, bool PHL = false>
int launch_igemm2d_t(const IgemmParams& pin, hipStream_t st) {
    IgemmParams p = pin;
    if (!p.out1) p.out0_c = p.N;
    constexpr int PITCH = CPS * 16 + 16, NT = NFRAG * 16;
    constexpr int PD = KS == 3 ? 15 * SA + 3 : KS == 2 ? 17 : 16;
    constexpr int MAXTAP = KS == 3 ? 9 : KS == 2 ? (PHL ? 9 : 4) : 1;
    constexpr int TAPPAD = (CPS == 2) ? ((MAXTAP + 1) & ~1) : MAXTAP;
    const int nz = PHL ? 1 : p.nphase;
    {
        // tiles per workgroup (consecutive along x): about eight workgroups per CU stay in flight
        const long long wgs = (long long)p.nb * p.ntile[1] * p.ntile[2] * (p.N / NT) * nz;
        int want = (int)(wgs / 2048);
        if (want > 8) want = 8;
        int tpw = 1;
        for (int t = 1; t <= want; ++t) if (p.ntile[2] % t == 0) tpw = t;
        p.tpw = tpw;
    }
    constexpr int WPITCH = CPS == 2 ? CPS * 16 : PITCH;     // unpadded, swizzled weight rows (see the kernel)
    size_t lds = (size_t)PD * PD * PITCH + (size_t)TAPPAD * NT * WPITCH;
    if (lds < (size_t)NT * 2 * 4 * 4) lds = NT * 2 * 4 * 4;
    auto kern = igemm2d_kernel<TR, CPS, NFRAG, KS, SA, PHL>;
    if (int r = set_lds(kern, lds)) return r;
    dim3 grid(p.nb * p.ntile[1] * (p.ntile[2] / p.tpw), p.N / NT, nz);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("igemm2d launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

template <class TR, int NT, int FF, int SF, int WF, int WS>
int launch_wgrad_t(const WgradParams& pin, hipStream_t st) {
    WgradParams p = pin;
    p.mg_pd2 = magic_of(p.pd[2]); p.mg_pd12 = magic_of(p.pd[1] * p.pd[2]);
    constexpr int FB = 16 * FF * WF, SB = 16 * SF * WS;
    constexpr int FPITCH = FB * TR::ESZ + 16, SPITCH = SB * TR::ESZ + 16;
    const int PP = p.pd[0] * p.pd[1] * p.pd[2];
    size_t lds = (size_t)256 * FPITCH + (size_t)PP * SPITCH;
    constexpr int WK = 4 / (WF * WS);
    if (WK > 1) {
        const size_t comb = (size_t)4 * (NT * FF * SF + FF) * 4 * 64 * sizeof(float);
        if (comb > lds) lds = comb;
    }
    auto kern = wgrad_kernel<TR, NT, FF, SF, WF, WS>;
    if (int r = set_lds(kern, lds)) return r;
    dim3 grid(p.nsplit, p.F / FB, (p.S / SB) * p.ntapg);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("wgrad launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

}  // namespace

static const size_t kLdsBudget = 150 * 1024;

static size_t igemm_lds_for(const IgemmParams& p, int cps, int nfrag) {
    const int PITCH = cps * 16 + 16;
    const int PP = p.pd[0] * p.pd[1] * p.pd[2];
    const int tapg_pad = (cps == 2) ? ((p.tapg + 1) & ~1) : p.tapg;
    return (size_t)PP * PITCH + (size_t)tapg_pad * nfrag * 16 * PITCH;
}

// Chooses the contraction chunk (CPS 16-byte pieces per pixel per stage) and the N tile
// (NFRAG 16-channel fragments) so that the LDS images fit.
void igemm_config(int dtype, const IgemmParams& p, int* cps, int* nfrag) {
    const int esz = dtype ? 2 : 4;
    const int kce4 = 64 / esz;
    int c = (p.K % kce4 == 0 && (p.in0_c >= p.K || p.in0_c % kce4 == 0)) ? 4 : 2;
    int nf = (p.N % 64 == 0) ? 4 : (p.N % 32 == 0) ? 2 : 1;
    static const bool no_igemm2d = getenv("MI_NO_IGEMM2D") != nullptr;     // read once: this runs on every launch
    if (!no_igemm2d && igemm2d_ks(p) == 3 && p.sa == 2) {
        // 33 x 33 patch: 32-byte chunks; with at most 64 output channels two 32-channel workgroups fit a CU
        c = 2;
        if (nf > 2 && p.N <= 64) nf = 2;
    }
    if (c == 4 && igemm_lds_for(p, c, nf) > kLdsBudget) c = 2;
    while (nf > 1 && igemm_lds_for(p, c, nf) > kLdsBudget) nf >>= 1;
    {
        // small problems: narrower channel tiles until there is about one workgroup per CU (2-D) or per four CUs
        // (3-D: a narrower tile stages the same 3-D patch, ten times the size of a 2-D one, once more per channel
        // block; measured on 96^3 patches: 1.846 -> 1.803 ms per step, the 2-D small-batch case prefers 256)
        int min_wgs = p.id[0] > 1 ? 64 : 256;
        static const int min_wgs_env = getenv("MI_IGEMM_MIN_WGS") ? atoi(getenv("MI_IGEMM_MIN_WGS")) : 0;
        if (min_wgs_env > 0) min_wgs = min_wgs_env;
        const long long tiles = (long long)p.nb * p.ntile[0] * p.ntile[1] * p.ntile[2] * p.nphase;
        while (nf > 1 && tiles * (p.N / (16 * nf)) < min_wgs) nf >>= 1;
    }
    *cps = c; *nfrag = nf;
}

size_t igemm_lds_bytes(int dtype, const IgemmParams& p) {
    int cps, nfrag;
    igemm_config(dtype, p, &cps, &nfrag);
    return igemm_lds_for(p, cps, nfrag);
}

bool conv2d_dma_enabled() {
    static const bool on = !(getenv("MI_NO_CONV_DMA") && atoi(getenv("MI_NO_CONV_DMA")) != 0);
    return on;
}

int launch_igemm(int dtype, const IgemmParams& p, hipStream_t st) {
    if (conv2d_dma_enabled() && conv2d_dma_applicable(dtype, p)) {
        MI_CHECK((long long)p.nb * p.od[0] * p.od[1] * p.od[2] < (1ll << 31) && (long long)p.nb * p.id[0] * p.id[1] * p.id[2] < (1ll << 31),
                 "tensor has too many pixels for 32-bit pixel indices");
        MI_CHECK(!p.out1 || (p.out0_c % 16 == 0 && p.out0_c > 0 && p.out0_c < p.N), "first-destination channel count must be a multiple of 16");
        return launch_conv2d_dma(dtype, p, st);
    }
    if (conv2d_dma_enabled() && conv3d_k16_applicable(dtype, p)) return launch_conv3d_k16(p, st);
    if (igemm_dma_applicable(dtype, p)) return launch_igemm_dma(dtype, p, st);
    int cps, nfrag;
    igemm_config(dtype, p, &cps, &nfrag);
    const int esz = dtype ? 2 : 4;
    MI_CHECK(p.K % (32 / esz) == 0, "contraction channels must be a multiple of 32 bytes");
    MI_CHECK(p.in0_c >= p.K || p.in0_c % (32 / esz) == 0, "first-source channel count must be a multiple of 32 bytes");
    MI_CHECK(p.N % 16 == 0, "output channels must be a multiple of 16");
    MI_CHECK(!p.out1 || (p.out0_c % 16 == 0 && p.out0_c > 0 && p.out0_c < p.N), "first-destination channel count must be a multiple of 16");
    MI_CHECK(p.tlog[0] + p.tlog[1] + p.tlog[2] == 8, "tile must hold 256 base pixels");
    MI_CHECK(p.tapg >= 1 && p.tapg <= 9, "tap group size");
    MI_CHECK(igemm_lds_for(p, cps, nfrag) <= 160 * 1024, "LDS images do not fit");
    MI_CHECK((long long)p.nb * p.od[0] * p.od[1] * p.od[2] < (1ll << 31) && (long long)p.nb * p.id[0] * p.id[1] * p.id[2] < (1ll << 31),
             "tensor has too many pixels for 32-bit pixel indices");
    MI_CHECK(p.ksplit <= 1 || (p.kpart != nullptr && p.N % 4 == 0 && p.N / 4 <= 256), "contraction split needs a partial-sum buffer");
    static const bool no_igemm2d = getenv("MI_NO_IGEMM2D") != nullptr;
    const int ks = (no_igemm2d || p.ksplit > 1) ? 0 : igemm2d_ks(p);
    int phl_max = 2;   // widest channel tile (in 16-channel fragments) that loops over the scatter phases in one workgroup
    static const int phl_env = getenv("MI_PHL_MAX") ? atoi(getenv("MI_PHL_MAX")) : -1;
    if (phl_env >= 0) phl_max = phl_env;
#define IG_CASE(TRT, C, NF)                                                  \
    if (cps == C && nfrag == NF) {                                           \
        if (ks == 3 && p.sa == 2 && C == 2) return launch_igemm2d_t<TRT, 2, NF, 3, 2>(p, st); \
        if (ks == 3 && p.sa == 1) return launch_igemm2d_t<TRT, C, NF, 3, 1>(p, st); \
        if (ks == 2 && C == 4 && NF <= phl_max) return launch_igemm2d_t<TRT, 4, (NF <= 2 ? NF : 1), 2, 1, true>(p, st); \
        if (ks == 2) return launch_igemm2d_t<TRT, C, NF, 2, 1>(p, st);       \
        if (ks == 1) return launch_igemm2d_t<TRT, C, NF, 1, 1>(p, st);       \
        return launch_igemm_t<TRT, C, NF>(p, st);                            \
    }
    if (dtype) {
        IG_CASE(BF16, 4, 4) IG_CASE(BF16, 4, 2) IG_CASE(BF16, 4, 1)
        IG_CASE(BF16, 2, 4) IG_CASE(BF16, 2, 2) IG_CASE(BF16, 2, 1)
    } else {
        IG_CASE(F32, 4, 4) IG_CASE(F32, 4, 2) IG_CASE(F32, 4, 1)
        IG_CASE(F32, 2, 4) IG_CASE(F32, 2, 2) IG_CASE(F32, 2, 1)
    }
#undef IG_CASE
    mi_set_error("launch_igemm: no kernel variant");
    return -1;
}

static size_t wgrad_lds_for(int dtype, const WgradParams& p, int FB, int SB) {
    const int esz = dtype ? 2 : 4;
    const int PP = p.pd[0] * p.pd[1] * p.pd[2];
    return (size_t)256 * (FB * esz + 16) + (size_t)PP * (SB * esz + 16);
}

// Weight-gradient block shape: FB x SB channels per workgroup (fragments per wave and wave
// arrangement follow), shrunk until the LDS images fit.
void wgrad_config(int dtype, const WgradParams& p, int* ff, int* sf, int* wf, int* wsn) {
    if (wgrad_thick_config(dtype, p, ff, sf, wf, wsn)) return;
    auto pick = [](int c, int maxblk) { int b = maxblk; while (b > 16 && c % b) b >>= 1; return b; };
    int maxb = dtype ? 64 : 32;
    static const int maxb_env = getenv("MI_WG_MAXB") ? atoi(getenv("MI_WG_MAXB")) : 0;
    if (maxb_env >= 16 && maxb_env < maxb) maxb = maxb_env;
    int FB = pick(p.F, maxb), SB = pick(p.S, maxb);
    if (p.f0_c < p.F) while (FB > 16 && p.f0_c % FB) FB >>= 1;
    while (wgrad_lds_for(dtype, p, FB, SB) > kLdsBudget && (SB > 16 || FB > 16)) {
        if (SB > 16 && SB >= FB) SB >>= 1; else if (FB > 16) FB >>= 1; else SB >>= 1;
    }
    *ff = FB >= 32 ? 2 : 1; *wf = FB / (16 * *ff);
    *sf = SB >= 32 ? 2 : 1; *wsn = SB / (16 * *sf);
}

// Chooses how many workgroups split the pixels of each (F block, S block, tap group): enough workgroups to
// fill the chip, never more than there are tiles.
void wgrad_choose_split(int dtype, WgradParams& p, int target) {
    int ff, sf, wf, wsn;
    p.nsplit = 1;
    wgrad_config(dtype, p, &ff, &sf, &wf, &wsn);
    const int blocks = (p.F / (16 * ff * wf)) * (p.S / (16 * sf * wsn)) * p.ntapg;
    // default: three quarters of the CUs.  A weight gradient runs next to the kernels of the critical chain; with
    // more workgroups it finishes sooner and slows them by more than it gains (measured 192 < 256 << 512).
    if (target <= 0) target = 192;
    static const int wgs_env = getenv("MI_WG_WGS") ? atoi(getenv("MI_WG_WGS")) : 0;
    if (wgs_env > 0) target = wgs_env;
    int nsplit = (target + blocks - 1) / blocks;
    if (nsplit > p.tiles_total) nsplit = p.tiles_total;
    if (nsplit < 1) nsplit = 1;
    p.nsplit = nsplit;
}

int wgrad_slabs(int dtype, const WgradParams& p) {
    (void)dtype;
    return p.nsplit;   // the pixel-split waves of a workgroup are combined before the slab is written
}

size_t wgrad_lds_bytes(int dtype, const WgradParams& p) {
    int ff, sf, wf, wsn;
    wgrad_config(dtype, p, &ff, &sf, &wf, &wsn);
    return wgrad_lds_for(dtype, p, 16 * ff * wf, 16 * sf * wsn);
}

int launch_wgrad(int dtype, const WgradParams& p, hipStream_t st) {
    int ff, sf, wf, wsn;
    wgrad_config(dtype, p, &ff, &sf, &wf, &wsn);
    MI_CHECK(p.F % 16 == 0 && p.S % 16 == 0, "channel counts must be multiples of 16");
    int tf, ts, twf, tws;
    const bool thick = wgrad_thick_config(dtype, p, &tf, &ts, &twf, &tws);
    MI_CHECK(thick || p.f0_c >= p.F || p.f0_c % (16 * ff * wf) == 0, "first-source channel count must be a multiple of the F block");
    MI_CHECK(p.tlog[0] + p.tlog[1] + p.tlog[2] == 8, "tile must hold 256 base pixels");
    MI_CHECK(wgrad_lds_for(dtype, p, 16 * ff * wf, 16 * sf * wsn) <= 160 * 1024, "LDS images do not fit");
    const int nt = p.ntaps_total / p.ntapg;
    MI_CHECK(nt == 9 || nt == 1, "taps per group must be 9 or 1");
    {
        static const bool dma = !(getenv("MI_NO_WGRAD_DMA") && atoi(getenv("MI_NO_WGRAD_DMA")) != 0);
        if (dma) {
            WgradParams q = p;
            int r = launch_wgrad_thick(dtype, q, ff, sf, wf, wsn, st);
            if (r <= 0) return r;
            r = launch_wgrad_dma(dtype, q, ff, sf, wf, wsn, st);
            if (r <= 0) return r;
        }
    }
#define WG_CASE(TRT, NTT, A, B, C, D) \
    if (nt == NTT && ff == A && sf == B && wf == C && wsn == D) return launch_wgrad_t<TRT, NTT, A, B, C, D>(p, st);
#define WG_ALL(TRT, NTT)                                                                  \
    WG_CASE(TRT, NTT, 1, 1, 1, 1) WG_CASE(TRT, NTT, 2, 1, 1, 1) WG_CASE(TRT, NTT, 1, 2, 1, 1) \
    WG_CASE(TRT, NTT, 2, 2, 1, 1) WG_CASE(TRT, NTT, 2, 1, 2, 1) WG_CASE(TRT, NTT, 1, 2, 1, 2) \
    WG_CASE(TRT, NTT, 2, 2, 2, 1) WG_CASE(TRT, NTT, 2, 2, 1, 2) WG_CASE(TRT, NTT, 2, 2, 2, 2)
    if (dtype) { WG_ALL(BF16, 9) WG_ALL(BF16, 1) }
    else { WG_ALL(F32, 9) WG_ALL(F32, 1) }
#undef WG_ALL
#undef WG_CASE
    mi_set_error("launch_wgrad: no kernel variant");
    return -1;
}

// Workgroups the forward / dgrad launch of `p` would use without a contraction split, and its contraction chunks
void igemm_parallelism(int dtype, const IgemmParams& p, long long* wgs, int* nkc) {
    int cps, nfrag;
    if (!igemm_dma_config(dtype, p, &cps, &nfrag)) igemm_config(dtype, p, &cps, &nfrag);
    *wgs = (long long)p.nb * p.ntile[0] * p.ntile[1] * p.ntile[2] * p.nphase * (p.N / (16 * nfrag));
    const int kce = cps * 16 / (dtype ? 2 : 4);
    *nkc = (p.K + kce - 1) / kce;
}

int launch_ksplit_finish(int dtype, const IgemmParams& pin, hipStream_t st) {
    IgemmParams p = pin;
    if (!p.out1) p.out0_c = p.N;
    MI_CHECK(!p.kpart_blocked, "blocked partial sums are read by the finish + norm launch");
    const int npix_img = p.od[0] * p.od[1] * p.od[2];
    const int rows = 256 / (p.N / 4);
    int gx = (npix_img + rows * 4 - 1) / (rows * 4);
    if (gx < 1) gx = 1;
    if (gx > 1024) gx = 1024;
    const size_t lds = 256 * 8 * sizeof(float);
    if (dtype) hipLaunchKernelGGL(ksplit_finish_kernel<BF16>, dim3(gx, p.nb), dim3(256), lds, st, p, npix_img);
    else hipLaunchKernelGGL(ksplit_finish_kernel<F32>, dim3(gx, p.nb), dim3(256), lds, st, p, npix_img);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("ksplit finish launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

// workgroups a descriptor with `total` output elements keeps busy: reduce_body gives a workgroup 64, 16, 4 or 1
// of them (few outputs with many slabs get more slab groups instead)
// Slab counts up to which the reduction goes through LDS tiles.  Measured with 8: 96^3 patches 1.503 -> 1.440 ms per
// step (reduction launches 37.9 -> 22.7 us), headline unchanged (its wide layers have 12 slabs; with 32 the launches
// are 6 us shorter and the step is not).  MI_REDUCE_TILED_SLABS sets the bound, MI_NO_REDUCE_TILED=1 is 0: never.
static int reduce_tiled_max_slabs() {
    static const int v = getenv("MI_NO_REDUCE_TILED") ? 0 : getenv("MI_REDUCE_TILED_SLABS") ? atoi(getenv("MI_REDUCE_TILED_SLABS")) : 8;
    return v;
}

int reduce_blocks_for(int total) {
    const int el = total >= 2048 ? 64 : total >= 256 ? 16 : total >= 16 ? 4 : 1;
    int g = (total + el - 1) / el;
    return g > 2048 ? 2048 : g < 1 ? 1 : g;
}

// total_blocks: sum of the descriptors' nblk (each set from reduce_blocks_for(), blk0 = running sum)
int launch_reduce(const ReduceTable& tab, int total_blocks, hipStream_t st) {
    if (tab.n <= 0 || total_blocks <= 0) return 0;
    static_assert(sizeof(ReduceTable) <= 4096 && sizeof(PackTable) <= 4096, "descriptor tables must fit the kernel argument segment");
    ReduceTable t = tab;
    for (int i = 0; i < t.n; ++i) t.d[i].tiled_max_slabs = reduce_tiled_max_slabs();
    hipLaunchKernelGGL(reduce_kernel, dim3(total_blocks), dim3(256), 0, st, t);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("reduce launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

// workgroups for a descriptor of `pairs` (n, k) pairs: one pair per thread
int pack_blocks_for(int pairs) {
    const int g = (pairs + 255) / 256;
    return g < 1 ? 1 : g > 1024 ? 1024 : g;
}

// total_blocks: sum of the descriptors' nblk (each set from pack_blocks_for(), blk0 = running sum)
int launch_pack(int dtype, const PackTable& tab, int total_blocks, hipStream_t st) {
    if (tab.n <= 0 || total_blocks <= 0) return 0;
    if (dtype) hipLaunchKernelGGL(pack_kernel<BF16>, dim3(total_blocks), dim3(256), 0, st, tab);
    else hipLaunchKernelGGL(pack_kernel<F32>, dim3(total_blocks), dim3(256), 0, st, tab);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("pack launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

int launch_reduce_one(const ReduceDesc& d, hipStream_t st) {
    const int gx = reduce_blocks_for(d.ntaps * d.Fcnt * d.S + d.Fcnt);
    ReduceDesc t = d;
    t.tiled_max_slabs = reduce_tiled_max_slabs();
    hipLaunchKernelGGL(reduce_one_kernel, dim3(gx), dim3(256), 0, st, t);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("reduce launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

int launch_pack_one(int dtype, const PackDesc& d, hipStream_t st) {
    int gx = (d.N * d.K + 255) / 256; if (gx > 256) gx = 256; if (gx < 1) gx = 1;
    if (dtype) hipLaunchKernelGGL(pack_one_kernel<BF16>, dim3(gx), dim3(256), 0, st, d);
    else hipLaunchKernelGGL(pack_one_kernel<F32>, dim3(gx), dim3(256), 0, st, d);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("pack launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}
