// Weight gradient with asynchronous staging (gfx950).
//
// Same contract, block shapes and arithmetic as wgrad_kernel (igemm.hip): partial[slab][tap][f][s] = sum over the
// slab's base pixels of Fm[b][f] * Sm[b * ss + off_tap][s].  What differs is how a tile (256 base pixels of the
// fixed operand, the matching patch of the shifted operand) reaches LDS: global_load ... lds into a ring of R
// slots, R - 1 tiles ahead of the one being multiplied and without a register in between.  The thin layers (16 and
// 32 channels) spend their time waiting for memory, not multiplying: with one tile in flight per workgroup, as the
// register-staged kernel has, a CU keeps far too few bytes in flight to reach the HBM rate; a ring of four does.
// Rows are unpadded (the DMA writes LDS lane-linearly); pixels outside the tensors come from a zero page.
#include "common.h"
#include "kparams.h"

namespace {

static __device__ __align__(256) unsigned char g_zero_page_w[256];

__device__ __forceinline__ void dma16w(const char* gsrc, char* lds_base) {
    // Inline assembly on purpose: hipcc drains every outstanding LDS-DMA it knows of (s_waitcnt vmcnt(0)) in front
    // of the next LDS read, which would undo the prefetch; loads it does not know of are waited for by the counted
    // s_waitcnt vmcnt(N) of the stage loops alone.  M0 holds the wave-uniform LDS destination and is restored.
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

template <int N>
__device__ __forceinline__ void wait_vmcnt_n() {
    static_assert(N >= 0, "count");
    if (N < 64) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(N < 64 ? N : 0) : "memory");
    else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
}

// Workgroup = 4 waves arranged WK x WF x WS as in wgrad_kernel; R = ring slots (2..4), NISW = patch DMA
// instructions per wave and tile (compile-time: the tile loop then has no dead branches and fits the scalar registers)
template <class TR, int NT, int FF, int SF, int WF, int WS, int NISW>
__global__ __launch_bounds__(256, 2) void wgrad_dma_kernel(const WgradParams p, int R, int nis, int slot_bytes) {
    constexpr int SMAXU = NISW;
    constexpr int ESZ = TR::ESZ;
    constexpr int WK = 4 / (WF * WS);
    constexpr int FB = 16 * FF * WF, SB = 16 * SF * WS;
    constexpr int FROW = FB * ESZ, SROW = SB * ESZ;          // unpadded LDS rows
    constexpr int LF = FROW / 16, LS = SROW / 16;            // lanes (16-byte pieces) per row
    constexpr int PPIF = 64 / LF, PPIS = 64 / LS;            // pixels per DMA wave-instruction
    constexpr int NIFW = LF;                                 // F instructions per wave and tile (4 * LF per tile)
    constexpr int FBYTES = 256 * FROW;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    const int ws = wave % WS, wf = (wave / WS) % WF, wk = wave / (WS * WF);
    const int fb0 = blockIdx.y * FB;
    const int sblk = blockIdx.z / p.ntapg, tgi = blockIdx.z % p.ntapg;
    const int sb0 = sblk * SB;
    const int pd12 = p.pd[1] * p.pd[2];
    const int PP = p.pd[0] * pd12;
    const int mx = (1 << p.tlog[2]) - 1, my = (1 << p.tlog[1]) - 1;
    constexpr int nisw = NISW;

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
    const int fpb = fpitch * ESZ, spb = p.s_pitch * ESZ;
    const char* const fbase = fsrc + (size_t)fcoff * ESZ;
    const char* const sbase = p.s + (size_t)(p.s_coff + sb0) * ESZ;
    const int tiles_per_img = p.ntile[0] * p.ntile[1] * p.ntile[2];
    const char* const zero = reinterpret_cast<const char*>(g_zero_page_w) + (lane % 4) * 16;

    // ---- per-lane DMA constants -------------------------------------------------------------------------------------
    int f_off[NIFW], f_zyx[NIFW];
#pragma unroll
    for (int u = 0; u < NIFW; ++u) {
        const int pix = (wave + 4 * u) * PPIF + lane / LF;           // tile pixel 0 .. 255
        const int tx = pix & mx, ty = (pix >> p.tlog[2]) & my, tz = pix >> (p.tlog[2] + p.tlog[1]);
        f_off[u] = ((tz * p.bd[1] + ty) * p.bd[2] + tx) * fpb + (lane % LF) * 16;
        f_zyx[u] = (tz << 20) | (ty << 10) | tx;
    }
    int s_off[SMAXU], s_zyx[SMAXU], s_dst[SMAXU];
    const bool patch_whole = PP % PPIS == 0;                         // no DMA instruction reaches past the patch
#pragma unroll
    for (int u = 0; u < SMAXU; ++u) {
        s_off[u] = 0; s_zyx[u] = -1; s_dst[u] = 0;
        if (u < nisw) {
            const int i = (wave + 4 * u) % nis;                      // past the end: a repeat, every wave issues nisw loads
            s_dst[u] = i * 1024;
            const int pix = i * PPIS + lane / LS;
            if (pix < PP) {
                const int pz = pix / pd12; const int r = pix - pz * pd12;
                const int py = r / p.pd[2]; const int px = r - py * p.pd[2];
                s_off[u] = ((pz * p.sd[1] + py) * p.sd[2] + px) * spb + (lane % LS) * 16;
                s_zyx[u] = (pz << 20) | (py << 10) | px;
            }
        }
    }

    // n / d through the reciprocal 2^32 / d + 1 (exact for the tile counts here, far below 2^16 * d); 0 stands for d == 1
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)d) + 1u; };
    auto fdiv = [](int n, unsigned mg) { return mg ? (int)__umulhi((unsigned)n, mg) : n; };
    const unsigned mg_img = magic(tiles_per_img), mg_x = magic(p.ntile[2]), mg_y = magic(p.ntile[1]);
    auto issue_tile = [&](int tile, int slot) {
        const int img = fdiv(tile, mg_img);
        int tr = tile - img * tiles_per_img;
        const int t1 = fdiv(tr, mg_x);
        const int txi = tr - t1 * p.ntile[2];
        const int tzi = fdiv(t1, mg_y);
        const int tyi = t1 - tzi * p.ntile[1];
        const int bz0 = tzi << p.tlog[0], by0 = tyi << p.tlog[1], bx0 = txi << p.tlog[2];
        char* dstF = smem + slot * slot_bytes;
        char* dstS = dstF + FBYTES;
        {
            const char* fb = fbase + (((long long)(img * p.bd[0] + bz0) * p.bd[1] + by0) * p.bd[2] + bx0) * fpb;
            const bool full = bz0 + (1 << p.tlog[0]) <= p.bd[0] && by0 + (1 << p.tlog[1]) <= p.bd[1] && bx0 + (1 << p.tlog[2]) <= p.bd[2];
            if (full) {
#pragma unroll
                for (int u = 0; u < NIFW; ++u) dma16w(fb + f_off[u], dstF + (wave + 4 * u) * 1024);
            } else {
#pragma unroll
                for (int u = 0; u < NIFW; ++u) {
                    const bool ok = bz0 + (f_zyx[u] >> 20) < p.bd[0] && by0 + ((f_zyx[u] >> 10) & 1023) < p.bd[1] && bx0 + (f_zyx[u] & 1023) < p.bd[2];
                    dma16w(ok ? fb + f_off[u] : zero, dstF + (wave + 4 * u) * 1024);
                }
            }
        }
        {
            const int iz0 = bz0 * p.ss + p.omin[0] + p.tg_zoff[tgi], iy0 = by0 * p.ss + p.omin[1], ix0 = bx0 * p.ss + p.omin[2];
            const char* sb = sbase + (((long long)(img * p.sd[0] + iz0) * p.sd[1] + iy0) * p.sd[2] + ix0) * spb;
            const bool interior = iz0 >= 0 && iy0 >= 0 && ix0 >= 0 && iz0 + p.pd[0] <= p.sd[0] && iy0 + p.pd[1] <= p.sd[1] && ix0 + p.pd[2] <= p.sd[2];
            if (interior && patch_whole) {
#pragma unroll
                for (int u = 0; u < SMAXU; ++u)
                    if (u < nisw) dma16w(sb + s_off[u], dstS + s_dst[u]);
            } else {
#pragma unroll
                for (int u = 0; u < SMAXU; ++u) {
                    if (u < nisw) {
                        bool ok = s_zyx[u] >= 0;
                        if (!interior && ok)
                            ok = (unsigned)(iz0 + (s_zyx[u] >> 20)) < (unsigned)p.sd[0] && (unsigned)(iy0 + ((s_zyx[u] >> 10) & 1023)) < (unsigned)p.sd[1] &&
                                 (unsigned)(ix0 + (s_zyx[u] & 1023)) < (unsigned)p.sd[2];
                        dma16w(ok ? sb + s_off[u] : zero, dstS + s_dst[u]);
                    }
                }
            }
        }
    };

    // ---- per-lane fragment addresses (bf16), the same for every tile ----------------------------------------------
    // MFMA contraction slot k = 8 g + 4 h + q (g = lane / 16, h = low / high transposed read, q = row of the 4-row
    // block) takes tile pixel 32 ks + 16 (g / 2) + 8 h + 4 (g % 2) + q, for both operands alike: the two 16-lane groups
    // of a 32-lane half then read rows 4 pixels apart (128 bytes apart at 32-byte rows) instead of 8 (a 2-way conflict).
    int fo[8 / WK][2], so[8 / WK][2], toffs[NT];
    if (TR::ESZ == 2) {
        const int q = lr >> 2, pp = lr & 3;
#pragma unroll
        for (int it = 0; it < 8 / WK; ++it)
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int pix = (wk + it * WK) * 32 + 16 * (g >> 1) + 8 * h + 4 * (g & 1) + q;
                const int tx = pix & mx, ty = (pix >> p.tlog[2]) & my, tz = pix >> (p.tlog[2] + p.tlog[1]);
                fo[it][h] = pix * FROW + (wf * FF * 16 + pp * 4) * 2;
                so[it][h] = (((tz * p.ss) * p.pd[1] + ty * p.ss) * p.pd[2] + tx * p.ss) * SROW + pp * 8 + (ws * SF * 16) * 2;
            }
#pragma unroll
        for (int t = 0; t < NT; ++t) toffs[t] = p.tap_off[t] * SROW;
    }

    // ---- ring: tile j of this workgroup is blockIdx.x + j * nsplit and lives in slot j % R -----------------------
    const int ntw = p.tiles_total > (int)blockIdx.x ? (p.tiles_total - 1 - (int)blockIdx.x) / p.nsplit + 1 : 0;
    constexpr int PT = NIFW + NISW;                      // this wave's loads per tile
    const int LA = R - 1;
    for (int j = 0; j < LA && j < ntw; ++j) issue_tile(blockIdx.x + j * p.nsplit, j);
    int slot = 0, slot_issue = LA % R;
    for (int j = 0; j < ntw; ++j) {
        // tiles j + 1 .. j + LA - 1 may stay in flight
        const int ahead = min(LA - 1, ntw - 1 - j);
        if (ahead >= 2) wait_vmcnt_n<2 * PT>(); else if (ahead == 1) wait_vmcnt_n<PT>(); else wait_vmcnt_n<0>();
        __builtin_amdgcn_s_barrier();                    // tile j has landed everywhere, tile j - 1 has been consumed
        asm volatile("" ::: "memory");
        if (j + LA < ntw) issue_tile(blockIdx.x + (j + LA) * p.nsplit, slot_issue);
        slot_issue = slot_issue + 1 == R ? 0 : slot_issue + 1;
        const char* smF = smem + slot * slot_bytes;
        const char* smS = smF + FBYTES;
        slot = slot + 1 == R ? 0 : slot + 1;
#pragma unroll
        for (int it = 0; it < 8 / WK; ++it) {
            if (TR::ESZ == 2) {
                uint4 afr[FF];
#pragma unroll
                for (int a = 0; a < FF; ++a) {
                    v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(smF + fo[it][0] + a * 32));
                    v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(smF + fo[it][1] + a * 32));
                    v8s f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    afr[a] = __builtin_bit_cast(uint4, f);
                }
                if (do_bias) {
                    const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
#pragma unroll
                    for (int a = 0; a < FF; ++a)
                        accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afr[a]), __builtin_bit_cast(v8bf, ones), accb[a], 0, 0, 0);
                }
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    const int toff = toffs[t];
                    uint4 bfr[SF];
#pragma unroll
                    for (int b = 0; b < SF; ++b) {
                        v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(smS + so[it][0] + toff + b * 32));
                        v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(smS + so[it][1] + toff + b * 32));
                        v8s f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        bfr[b] = __builtin_bit_cast(uint4, f);
                    }
#pragma unroll
                    for (int a = 0; a < FF; ++a)
#pragma unroll
                        for (int b = 0; b < SF; ++b)
                            acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afr[a]), __builtin_bit_cast(v8bf, bfr[b]), acc[t][a][b], 0, 0, 0);
                }
            } else {
                const int ks = wk + it * WK;
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    const int pix = ks * 32 + 4 * i + g;
                    const int tx = pix & mx, ty = (pix >> p.tlog[2]) & my, tz = pix >> (p.tlog[2] + p.tlog[1]);
                    const int soff = (((tz * p.ss) * p.pd[1] + ty * p.ss) * p.pd[2] + tx * p.ss) * SROW;
                    float av[FF];
#pragma unroll
                    for (int a = 0; a < FF; ++a) av[a] = *reinterpret_cast<const float*>(smF + pix * FROW + ((wf * FF + a) * 16 + lr) * 4);
                    if (do_bias) {
#pragma unroll
                        for (int a = 0; a < FF; ++a) accb[a] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], 1.0f, accb[a], 0, 0, 0);
                    }
#pragma unroll
                    for (int t = 0; t < NT; ++t) {
                        const int toff = p.tap_off[t] * SROW;
#pragma unroll
                        for (int b = 0; b < SF; ++b) {
                            const float bv = *reinterpret_cast<const float*>(smS + soff + toff + ((ws * SF + b) * 16 + lr) * 4);
#pragma unroll
                            for (int a = 0; a < FF; ++a) acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x4f32(av[a], bv, acc[t][a][b], 0, 0, 0);
                        }
                    }
                }
            }
        }
    }
    // ---- combine the WK pixel-split waves in a fixed order through LDS, then write one slab -------
    if (WK > 1) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);   // [WK][(NT*FF*SF + FF)*4][64 lanes]
        constexpr int NV = (NT * FF * SF + FF) * 4;
        const int sub = wave % (WF * WS);
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
                for (int j = 0; j < 4; ++j) p.partial_bias[slab * p.F + fb0 + (wf * FF + a) * 16 + 4 * g + j] = accb[a][j];
        }
    }
}

template <class TR, int NT, int FF, int SF, int WF, int WS, int NISW>
int launch_wgrad_dma_t(const WgradParams& p, int nis, hipStream_t st) {
    constexpr int ESZ = TR::ESZ;
    constexpr int FB = 16 * FF * WF, WK = 4 / (WF * WS);
    constexpr int FROW = FB * ESZ;
    const int slot_bytes = 256 * FROW + nis * 1024;
    int R = (int)((150 * 1024) / slot_bytes);
    if (R > 4) R = 4;                                                   // two tiles in flight across a barrier at most
    if (R < 2) return 1;
    constexpr int PT = FROW / 16 + NISW;
    while (R > 2 && (R - 2) * PT > 60) --R;                             // vmcnt is a 6-bit counter
    size_t lds = (size_t)R * slot_bytes;
    if (WK > 1) {
        const size_t comb = (size_t)4 * (NT * FF * SF + FF) * 4 * 64 * sizeof(float);
        if (comb > lds) lds = comb;
    }
    if (lds > 160 * 1024) return 1;
    auto kern = wgrad_dma_kernel<TR, NT, FF, SF, WF, WS, NISW>;
    if (lds > 48 * 1024) {
        hipError_t e = mi_func_set_lds(reinterpret_cast<const void*>(kern), (int)lds);
        if (e != hipSuccess) { mi_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return -2; }
    }
    dim3 grid(p.nsplit, p.F / FB, (p.S / (16 * SF * WS)) * p.ntapg);
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, st, p, R, nis, slot_bytes);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("wgrad_dma launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

}  // namespace

// Returns 0 when launched, 1 when this kernel does not serve the shape (the caller falls back), < 0 on error.
int launch_wgrad_dma(int dtype, const WgradParams& p, int ff, int sf, int wf, int wsn, hipStream_t st) {
    const int nt = p.ntaps_total / p.ntapg;
    if (nt != 9 && nt != 1) return 1;
    for (int d = 0; d < 3; ++d)
        if (p.bd[d] > 1023 || p.sd[d] > 1023 || p.pd[d] > 1023) return 1;   // coordinates are packed in 10 bits
    if ((long long)p.sd[1] * p.sd[2] * p.pd[0] * p.s_pitch * 4 >= (1ll << 31) || (long long)p.bd[1] * p.bd[2] * 256 * (p.f0_pitch > p.f1_pitch ? p.f0_pitch : p.f1_pitch) * 4 >= (1ll << 31)) return 1;
    // thin blocks only (16 or 32 channels a side): they wait for memory; the wide blocks are bound by their MFMA /
    // LDS work and keep the register-staged kernel, whose padded rows suit them better
    if (nt != 9 || wf != 1 || wsn != 1) return 1;
    const int esz = dtype ? 2 : 4;
    const int srow = 16 * sf * esz, ppis = 64 / (srow / 16);
    const int PP = p.pd[0] * p.pd[1] * p.pd[2];
    const int nis = (PP + ppis - 1) / ppis, nisw = (nis + 3) / 4;
#define WG_N(TRT, A, B, NW) if (nisw == NW) return launch_wgrad_dma_t<TRT, 9, A, B, 1, 1, NW>(p, nis, st);
#define WG_CASE(TRT, A, B) \
    if (ff == A && sf == B) { WG_N(TRT, A, B, 3) WG_N(TRT, A, B, 6) WG_N(TRT, A, B, 9) WG_N(TRT, A, B, 11) WG_N(TRT, A, B, 18) return 1; }
    if (dtype) { WG_CASE(BF16, 1, 1) WG_CASE(BF16, 2, 1) WG_CASE(BF16, 1, 2) WG_CASE(BF16, 2, 2) }
    else { WG_CASE(F32, 1, 1) WG_CASE(F32, 2, 1) WG_CASE(F32, 1, 2) WG_CASE(F32, 2, 2) }
#undef WG_CASE
#undef WG_N
    return 1;
}
