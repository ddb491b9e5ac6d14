// Weight gradient of the wide 2-D layers with asynchronous staging (gfx950, bf16).
//
// Same contract as wgrad_kernel (igemm.hip): partial[slab][tap][f][s] = sum over the slab's base pixels of
// Fm[b][f] * Sm[b + off_tap][s], 3 x 3 taps, unit stride, 16 x 16 base-pixel tiles; a workgroup owns a 64 x 64 channel
// block (four waves, 32 x 32 channels each, all nine taps in registers).  The register-staged kernel spends a third
// of a tile's time computing addresses, issuing loads and writing LDS, holds 80 registers of prefetched data across
// its multiply loop (372 registers: one workgroup per CU) and has nothing to overlap those phases with.  Here
//  * a stage is half a tile (8 x 16 base pixels of the fixed operand, the 10 x 18 patch of the shifted one: 39 KB),
//    two stages per workgroup, two workgroups per CU;
//  * both operands go global -> LDS by global_load_lds (no registers in between), the next stage lands while the
//    current one is multiplied; the per-lane source offsets are linear in the instruction number (fixed operand) or
//    six precomputed values (patch), so no address arithmetic remains in the stage loop;
//  * LDS rows are unpadded 128-byte pixel rows whose 32-byte granules are permuted by the pixel's column
//    (granule j of column x sits in slot j ^ ((x >> 1) & 3)): a transposed read of a 32-lane half touches 8
//    consecutive pixels, whose rows start on alternating bank halves; the permutation puts the four rows of a half
//    on four different bank octets.  The DMA applies it to the source address;
//  * every fragment address is one per-lane register plus an immediate (k-step, tap row), 16 registers in all.
#include "common.h"
#include "kparams.h"

namespace {

static __device__ __align__(256) unsigned char g_zero_page_t[256];

__device__ __forceinline__ void dma16t(const char* gsrc, char* lds_base) {
    // inline assembly: the compiler would drain every LDS-DMA it knows of in front of the next LDS read
    const unsigned dst = __builtin_amdgcn_readfirstlane((unsigned)(uintptr_t)(__attribute__((address_space(3))) char*)lds_base);
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0"
                 : "=&s"(keep) : "v"(gsrc), "s"(dst) : "memory");
}

__device__ __forceinline__ int gswz(int x) { return (x >> 1) & 3; }

__global__ __launch_bounds__(256, 2) void wgrad_thick_kernel(const WgradParams p) {
    constexpr int ROWB = 128;                        // 64 bf16 channels per LDS row, both operands
    constexpr int PD = 18;
    constexpr int FPX = 128, SPX = 10 * PD;          // pixels per stage
    constexpr int NIF = FPX / 8, NIS = (SPX + 7) / 8;  // DMA wave-instructions per stage: 16 and 23 (8 pixels each)
    constexpr int FBYTES = NIF * 1024, SBYTES = NIS * 1024;
    constexpr int SLOT = FBYTES + SBYTES;
    constexpr int NUF = NIF / 4, NUS = (NIS + 3) / 4;  // per wave: 4 and 6
    constexpr int NT = 9;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    const int ws = wave & 1, wf = wave >> 1;
    const int fb0 = blockIdx.y * 64, sb0 = blockIdx.z * 64;
    const int H = p.bd[1], W = p.bd[2];              // unit stride: both operands have the base grid's extent

    v4f acc[NT][2][2];
    v4f accb[2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b) acc[t][a][b] = v4f{0.f, 0.f, 0.f, 0.f};
    accb[0] = v4f{0.f, 0.f, 0.f, 0.f}; accb[1] = v4f{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = (ws == 0) && (blockIdx.z == 0);

    const int spb = p.s_pitch * 2;
    const char* const sbase = p.s + (size_t)(p.s_coff + sb0) * 2;
    const char* const zero = reinterpret_cast<const char*>(g_zero_page_t) + (lane & 7) * 16;

    // ---- per-lane DMA constants ------------------------------------------------------------------------------------
    // eight lanes fill one 128-byte row; LDS slot sl of the row takes source chunk (((sl >> 1) ^ gswz(column)) << 1) | (sl & 1)
    const int sl = lane & 7;
    const int f_col = 8 * (wave & 1) + (lane >> 3);  // instruction wave + 4u: stage row 2u + (wave >> 1), this column
    const int f_row0 = wave >> 1;
    // the lane's 8 channels come from the first or the second source tensor of the fixed operand (a block may span both)
    const int f_ch = fb0 + ((((sl >> 1) ^ gswz(f_col)) << 1) | (sl & 1)) * 8;
    const bool f_first = f_ch < p.f0_c;
    const int fpb = (f_first ? p.f0_pitch : p.f1_pitch) * 2;
    const char* const fbase = f_first ? p.f0 + (size_t)(p.f0_coff + f_ch) * 2 : p.f1 + (size_t)(p.f1_coff + f_ch - p.f0_c) * 2;
    const int f_off0 = (f_row0 * W + f_col) * fpb;
    const int f_ustep = 2 * W * fpb;
    int s_off[NUS], s_yx[NUS];
#pragma unroll
    for (int u = 0; u < NUS; ++u) {
        const int i = wave + 4 * u;
        const int ps = i * 8 + (lane >> 3);
        s_off[u] = 0; s_yx[u] = -1;
        if (i < NIS && ps < SPX) {
            const int py = ps / PD, px = ps - py * PD;
            s_off[u] = (py * W + px) * spb + ((((sl >> 1) ^ gswz(px)) << 1) | (sl & 1)) * 16;
            s_yx[u] = (py << 8) | px;
        }
    }

    const int tiles_per_img = p.ntile[1] * p.ntile[2];
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)d) + 1u; };
    auto fdiv = [](int n, unsigned mg) { return mg ? (int)__umulhi((unsigned)n, mg) : n; };
    const unsigned mg_img = magic(tiles_per_img), mg_x = magic(p.ntile[2]);

    // stage j of this workgroup: half (j & 1) of tile blockIdx.x + (j >> 1) * nsplit, in slot j & 1
    auto issue_stage = [&](int j) {
        const int tile = blockIdx.x + (j >> 1) * p.nsplit;
        const int img = fdiv(tile, mg_img);
        const int tr = tile - img * tiles_per_img;
        const int tyi = fdiv(tr, mg_x);
        const int txi = tr - tyi * p.ntile[2];
        const int by0 = (tyi << 4) + 8 * (j & 1), bx0 = txi << 4;
        char* dstF = smem + (j & 1) * SLOT;
        char* dstS = dstF + FBYTES;
        {
            const char* fb = fbase + (((long long)img * H + by0) * W + bx0) * fpb;
            const bool colok = bx0 + f_col < W;
#pragma unroll
            for (int u = 0; u < NUF; ++u) {
                const bool ok = colok && by0 + f_row0 + 2 * u < H;
                dma16t(ok ? fb + f_off0 + u * f_ustep : zero, dstF + (wave + 4 * u) * 1024);
            }
        }
        {
            const int iy0 = by0 - 1, ix0 = bx0 - 1;
            const char* sb = sbase + (((long long)img * H + iy0) * W + ix0) * spb;
#pragma unroll
            for (int u = 0; u < NUS; ++u) {
                if (wave + 4 * u < NIS) {
                    const bool ok = s_yx[u] >= 0 && (unsigned)(iy0 + (s_yx[u] >> 8)) < (unsigned)H && (unsigned)(ix0 + (s_yx[u] & 255)) < (unsigned)W;
                    dma16t(ok ? sb + s_off[u] : zero, dstS + (wave + 4 * u) * 1024);
                }
            }
        }
    };

    // ---- per-lane fragment addresses ---------------------------------------------------------------------------------
    // contraction slot 8 g + 4 h + q of a k-step (32 pixels = two 16-pixel rows of the stage) takes pixel
    // v = 16 (g / 2) + 8 h + 4 (g % 2) + q: a 32-lane half reads 8 consecutive pixels of one row
    const int q = lr >> 2, pp = lr & 3;
    int fa0[2][2], sa0[2][3][2];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int v = 16 * (g >> 1) + 8 * h + 4 * (g & 1) + q;
        const int r = v >> 4, c = v & 15;
#pragma unroll
        for (int a = 0; a < 2; ++a) fa0[h][a] = v * ROWB + (((wf * 2 + a) ^ gswz(c)) * 32) + pp * 8;
#pragma unroll
        for (int d = 0; d < 3; ++d)
#pragma unroll
            for (int b = 0; b < 2; ++b) sa0[h][d][b] = (r * PD + c + d) * ROWB + (((ws * 2 + b) ^ gswz(c + d)) * 32) + pp * 8 + FBYTES;
    }

    const int ntw = p.tiles_total > (int)blockIdx.x ? (p.tiles_total - 1 - (int)blockIdx.x) / p.nsplit + 1 : 0;
    const int nstage = 2 * ntw;
    if (nstage > 0) issue_stage(0);
    for (int j = 0; j < nstage; ++j) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // stage j has landed everywhere, stage j - 1 has been consumed
        asm volatile("" ::: "memory");
        if (j + 1 < nstage) issue_stage(j + 1);
        const char* sm = smem + (j & 1) * SLOT;
        // One software-pipelined stream of 4 k-steps x 9 taps: the fixed-operand fragments of the next k-step and the
        // shifted-operand fragments of the next tap are requested before the MFMAs of the current tap are issued (two
        // register sets each).  With one workgroup per CU (the side-stream launches are sized that way) a wave has
        // nobody to hide its LDS latency behind; a single fragment set made every tap wait for its own reads.
        auto load_a = [&](int ks, uint4 (&afr)[2]) {
#pragma unroll
            for (int a = 0; a < 2; ++a) {
                v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(sm + fa0[0][a] + ks * 32 * ROWB));
                v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(sm + fa0[1][a] + ks * 32 * ROWB));
                v8s f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                afr[a] = __builtin_bit_cast(uint4, f);
            }
        };
        auto load_b = [&](int ks, int t, uint4 (&bfr)[2]) {
            const int dy = t / 3, dx = t % 3;
#pragma unroll
            for (int b = 0; b < 2; ++b) {
                v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(sm + sa0[0][dx][b] + (2 * ks + dy) * PD * ROWB));
                v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(sm + sa0[1][dx][b] + (2 * ks + dy) * PD * ROWB));
                v8s f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                bfr[b] = __builtin_bit_cast(uint4, f);
            }
        };
        uint4 aq[2][2], bq[2][2];
        load_a(0, aq[0]);
        load_b(0, 0, bq[0]);
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                const int step = ks * NT + t;
                if (t + 1 < NT) load_b(ks, t + 1, bq[(step + 1) & 1]);
                else if (ks + 1 < 4) { load_a(ks + 1, aq[(ks + 1) & 1]); load_b(ks + 1, 0, bq[(step + 1) & 1]); }
                __builtin_amdgcn_sched_barrier(0);       // keeps the requests of the next step in front of the MFMAs of this one
                if (t == 0 && do_bias) {
                    const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
#pragma unroll
                    for (int a = 0; a < 2; ++a)
                        accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, aq[ks & 1][a]), __builtin_bit_cast(v8bf, ones), accb[a], 0, 0, 0);
                }
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b)
                        acc[t][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, aq[ks & 1][a]), __builtin_bit_cast(v8bf, bq[step & 1][b]), acc[t][a][b], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }

    // ---- one slab per workgroup ----------------------------------------------------------------------------------------
    const size_t slab = (size_t)blockIdx.x;
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const size_t tbase = (slab * p.ntaps_total + t) * p.F;
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < 2; ++b)
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    const int f = fb0 + (wf * 2 + a) * 16 + 4 * g + j;
                    const int s = sb0 + (ws * 2 + b) * 16 + lr;
                    p.partial[(tbase + f) * p.S + s] = acc[t][a][b][j];
                }
    }
    if (do_bias && lr == 0) {
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) p.partial_bias[slab * p.F + fb0 + (wf * 2 + a) * 16 + 4 * g + j] = accb[a][j];
    }
}

// ---- the same for gather stride 2 (stride-2 convolution, transposed convolution), 64 x 32 channel blocks ----------------
// Base pixel b meets shifted pixels 2 b - 1 + k.  A tap reads one polyphase component of the shifted operand
// (component (ry, rx) = pixels (2 i - ry, 2 j - rx)), and inside a component the gather has unit stride, so the
// fragment reads look like the unit-stride kernel's.  A half tile is multiplied in three sub-stages that share its
// fixed-operand image: components (0,0) + (0,1) (taps (1,1); (1,0) (1,2)), (1,0) (taps (0,1) (2,1)) and (1,1) (taps
// (0,0) (0,2) (2,0) (2,2)); a component patch is 9 x 17 pixels of 64 bytes (32 channels), staged by ten DMA
// instructions whose source addresses step two pixels.  Fixed-operand images and component patches are each
// double-buffered (72 KB: two workgroups per CU).  Four waves = two 32-channel halves of the fixed operand x two
// halves of the k-steps; the k-halves are added through LDS at the end, nine taps in two rounds.
__device__ __forceinline__ int gswz64(int x) { return (x >> 2) & 1; }

template <int SF>   // 16-channel fragments of the shifted operand per workgroup: 2 (64-byte rows) or 1 (32-byte rows, unpermuted)
__global__ __launch_bounds__(256, 2) void wgrad_thick_s2_kernel(const WgradParams p) {
    constexpr int ROWF = 128, ROWS = 32 * SF;
    constexpr int PD = 17, CPX = 9 * PD;             // component patch: 153 pixels
    constexpr int LPP = ROWS / 16, PPI = 64 / LPP;   // lanes per patch pixel, pixels per DMA instruction
    constexpr int NIF = 16, NIC = (CPX + PPI - 1) / PPI;   // DMA wave-instructions: fixed image (8 pixels each), component: 10 or 5
    constexpr int FBYTES = NIF * 1024, CBYTES = NIC * 1024;
    constexpr int NUF = NIF / 4, NUC = (NIC + 3) / 4;  // per wave: 4 and 3
    constexpr int NT = 9;
    extern __shared__ __attribute__((aligned(1024))) char smem[];
    char* const smF = smem;                          // two fixed-operand images
    char* const smS = smem + 2 * FBYTES;             // ring of four component patches

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    const int wf = wave & 1, wk = wave >> 1;
    const int fb0 = blockIdx.y * 64, sb0 = blockIdx.z * 16 * SF;
    const int H = p.bd[1], W = p.bd[2], HS = p.sd[1], WS = p.sd[2];

    v4f acc[NT][2][SF];
    v4f accb[2];
#pragma unroll
    for (int t = 0; t < NT; ++t)
#pragma unroll
        for (int a = 0; a < 2; ++a)
#pragma unroll
            for (int b = 0; b < SF; ++b) acc[t][a][b] = v4f{0.f, 0.f, 0.f, 0.f};
    accb[0] = v4f{0.f, 0.f, 0.f, 0.f}; accb[1] = v4f{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = blockIdx.z == 0;

    const int spb = p.s_pitch * 2;
    const char* const sbase = p.s + (size_t)(p.s_coff + sb0) * 2;
    const char* const zero = reinterpret_cast<const char*>(g_zero_page_t) + (lane & 7) * 16;

    // ---- per-lane DMA constants ------------------------------------------------------------------------------------
    const int slf = lane & 7;
    const int f_col = 8 * (wave & 1) + (lane >> 3);
    const int f_row0 = wave >> 1;
    const int f_ch = fb0 + ((((slf >> 1) ^ gswz(f_col)) << 1) | (slf & 1)) * 8;
    const bool f_first = f_ch < p.f0_c;
    const int fpb = (f_first ? p.f0_pitch : p.f1_pitch) * 2;
    const char* const fbase = f_first ? p.f0 + (size_t)(p.f0_coff + f_ch) * 2 : p.f1 + (size_t)(p.f1_coff + f_ch - p.f0_c) * 2;
    const int f_off0 = (f_row0 * W + f_col) * fpb;
    const int f_ustep = 2 * W * fpb;
    const int sls = lane % LPP;                      // LPP lanes fill one row
    int c_off[NUC], c_yx[NUC];
#pragma unroll
    for (int u = 0; u < NUC; ++u) {
        const int i = (wave + 4 * u) % NIC;
        const int ps = i * PPI + lane / LPP;
        c_off[u] = 0; c_yx[u] = -1;
        if (ps < CPX) {
            const int py = ps / PD, px = ps - py * PD;
            c_off[u] = (2 * py * WS + 2 * px) * spb + (SF == 2 ? ((((sls >> 1) ^ gswz64(px)) << 1) | (sls & 1)) : sls) * 16;
            c_yx[u] = (py << 8) | px;
        }
    }

    const int tiles_per_img = p.ntile[1] * p.ntile[2];
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)d) + 1u; };
    auto fdiv = [](int n, unsigned mg) { return mg ? (int)__umulhi((unsigned)n, mg) : n; };
    const unsigned mg_img = magic(tiles_per_img), mg_x = magic(p.ntile[2]);

    // Sub-stage m = 4 * hs + k multiplies component k of half tile hs (half (hs & 1) of tile blockIdx.x + (hs >> 1) *
    // nsplit); its patch lives in ring slot k, the half tile's fixed image (brought along with component 0) in image
    // slot hs & 1.  Loads run three sub-stages ahead.
    int q_m = 0, q_k = 0, q_hs = 0, q_img = 0, q_by0 = 0, q_bx0 = 0;
    const int ntw = p.tiles_total > (int)blockIdx.x ? (p.tiles_total - 1 - (int)blockIdx.x) / p.nsplit + 1 : 0;
    const int nhs = 2 * ntw, nsub = 4 * nhs;
    auto issue_next = [&]() {
        if (q_m < nsub) {
            if (q_k == 0) {
                const int tile = blockIdx.x + (q_hs >> 1) * p.nsplit;
                q_img = fdiv(tile, mg_img);
                const int tr = tile - q_img * tiles_per_img;
                const int tyi = fdiv(tr, mg_x);
                const int txi = tr - tyi * p.ntile[2];
                q_by0 = (tyi << 4) + 8 * (q_hs & 1); q_bx0 = txi << 4;
                char* dstF = smF + (q_hs & 1) * FBYTES;
                const char* fb = fbase + (((long long)q_img * H + q_by0) * W + q_bx0) * fpb;
                const bool colok = q_bx0 + f_col < W;
#pragma unroll
                for (int u = 0; u < NUF; ++u) {
                    const bool ok = colok && q_by0 + f_row0 + 2 * u < H;
                    dma16t(ok ? fb + f_off0 + u * f_ustep : zero, dstF + (wave + 4 * u) * 1024);
                }
            }
            const int ry = q_k >> 1, rx = q_k & 1;
            const int iy0 = 2 * q_by0 - ry, ix0 = 2 * q_bx0 - rx;
            const char* sb = sbase + (((long long)q_img * HS + iy0) * WS + ix0) * spb;
            char* dstS = smS + q_k * CBYTES;
#pragma unroll
            for (int u = 0; u < NUC; ++u) {
                const int i = (wave + 4 * u) % NIC;      // past the end: a harmless repeat, so every wave issues NUC loads
                const bool ok = c_yx[u] >= 0 && (unsigned)(iy0 + 2 * (c_yx[u] >> 8)) < (unsigned)HS && (unsigned)(ix0 + 2 * (c_yx[u] & 255)) < (unsigned)WS;
                dma16t(ok ? sb + c_off[u] : zero, dstS + i * 1024);
            }
        }
        ++q_m;
        if (++q_k == 4) { q_k = 0; ++q_hs; }
    };

    // ---- per-lane fragment addresses ---------------------------------------------------------------------------------
    const int q = lr >> 2, pp = lr & 3;
    int fa0[2][2], sa0[2][2][SF];                     // [half of the k-step][..]: fixed [fragment]; shifted [column offset][fragment]
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int v = 16 * (g >> 1) + 8 * h + 4 * (g & 1) + q;
        const int r = v >> 4, c = v & 15;
#pragma unroll
        for (int a = 0; a < 2; ++a) fa0[h][a] = v * ROWF + (((wf * 2 + a) ^ gswz(c)) * 32) + pp * 8;
#pragma unroll
        for (int d = 0; d < 2; ++d)
#pragma unroll
            for (int b = 0; b < SF; ++b) sa0[h][d][b] = (r * PD + c + d) * ROWS + (SF == 2 ? ((b ^ gswz64(c + d)) * 32) : 0) + pp * 8;
    }

    // one tap: the two shifted-operand fragments of the component patch in ring slot CS at row / column offset
    // (IY, IX), four MFMAs
#define WG3_TAP(T, CS, IY, IX)                                                                                          \
    do {                                                                                                                \
        const uint4* afr = afrk[kk];                                                                                    \
        uint4 bfr_[SF];                                                                                                 \
        _Pragma("unroll") for (int b = 0; b < SF; ++b) {                                                                 \
            const char* base_ = smS + (CS) * CBYTES + (2 * ks + (IY)) * PD * ROWS;                                      \
            v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(base_ + sa0[0][IX][b])); \
            v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(base_ + sa0[1][IX][b])); \
            v8s f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};                                           \
            bfr_[b] = __builtin_bit_cast(uint4, f);                                                                     \
        }                                                                                                               \
        _Pragma("unroll") for (int a = 0; a < 2; ++a)                                                                   \
            _Pragma("unroll") for (int b = 0; b < SF; ++b)                                                              \
                acc[T][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afr[a]), __builtin_bit_cast(v8bf, bfr_[b]), acc[T][a][b], 0, 0, 0); \
    } while (0)
    constexpr int LC = NUC, LF = NUC + NUF;          // this wave's loads of a sub-stage without / with the fixed image
#pragma unroll
    for (int j = 0; j < 3; ++j) issue_next();
    for (int hs = 0; hs < nhs; ++hs) {
        const char* sF = smF + (hs & 1) * FBYTES;
        uint4 afrk[2][2];
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            // the loads of the next two sub-stages may stay in flight (the third is issued below)
            const int rem = nsub - 1 - (4 * hs + k);
            if (rem >= 2) {
                if (k == 0 || k == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * LC) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LC + LF) : "memory");
            } else if (rem == 1) {
                if (k == 3) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LF) : "memory");
                else asm volatile("s_waitcnt vmcnt(%0)" ::"n"(LC) : "memory");
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            __builtin_amdgcn_s_barrier();            // sub-stage 4 hs + k has landed everywhere, the one before has been consumed
            asm volatile("" ::: "memory");
            issue_next();
            if (k == 0) {
                // the fixed-operand fragments of the wave's two k-steps are read once per half tile (its image stays in
                // its slot for all four sub-stages) and kept in registers: 8 instead of 32 transposed reads
#pragma unroll
                for (int kk = 0; kk < 2; ++kk)
#pragma unroll
                    for (int a = 0; a < 2; ++a) {
                        const int ks = 2 * kk + wk;
                        v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(sF + fa0[0][a] + ks * 32 * ROWF));
                        v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(sF + fa0[1][a] + ks * 32 * ROWF));
                        v8s f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        afrk[kk][a] = __builtin_bit_cast(uint4, f);
                    }
            }
#pragma unroll
            for (int kk = 0; kk < 2; ++kk) {
                const int ks = 2 * kk + wk;          // this wave's k-steps of the half tile
                if (k == 0) {
                    if (do_bias) {
                        const uint4 ones = make_uint4(0x3f803f80u, 0x3f803f80u, 0x3f803f80u, 0x3f803f80u);
#pragma unroll
                        for (int a = 0; a < 2; ++a)
                            accb[a] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afrk[kk][a]), __builtin_bit_cast(v8bf, ones), accb[a], 0, 0, 0);
                    }
                    WG3_TAP(4, 0, 0, 0);             // component (0,0): tap (1,1)
                } else if (k == 1) {
                    WG3_TAP(3, 1, 0, 0);             // component (0,1): taps (1,0), (1,2)
                    WG3_TAP(5, 1, 0, 1);
                } else if (k == 2) {
                    WG3_TAP(1, 2, 0, 0);             // component (1,0): taps (0,1), (2,1)
                    WG3_TAP(7, 2, 1, 0);
                } else {
                    WG3_TAP(0, 3, 0, 0);             // component (1,1): taps (0,0), (0,2), (2,0), (2,2)
                    WG3_TAP(2, 3, 0, 1);
                    WG3_TAP(6, 3, 1, 0);
                    WG3_TAP(8, 3, 1, 1);
                }
            }
        }
    }
#undef WG3_TAP

    // ---- add the two k-halves through LDS (taps 0..4, then taps 5..8 and the bias column), then one slab ---------------
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    float* red = reinterpret_cast<float*>(smem);     // [wf][value][64 lanes]
#pragma unroll
    for (int round = 0; round < 2; ++round) {
        const int t0 = round ? 5 : 0, t1 = round ? NT : 5;
        __syncthreads();
        if (wk == 1) {
            float* mine = red + (size_t)wf * (40 * SF + 8) * 64 + lane;
            int v = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t >= t0 && t < t1) {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < SF; ++b)
#pragma unroll
                            for (int j = 0; j < 4; ++j) mine[(v++) * 64] = acc[t][a][b][j];
                }
            if (round == 1) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) mine[(v++) * 64] = accb[a][j];
            }
        }
        __syncthreads();
        if (wk == 0) {
            const float* theirs = red + (size_t)wf * (40 * SF + 8) * 64 + lane;
            int v = 0;
#pragma unroll
            for (int t = 0; t < NT; ++t)
                if (t >= t0 && t < t1) {
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < SF; ++b)
#pragma unroll
                            for (int j = 0; j < 4; ++j) acc[t][a][b][j] += theirs[(v++) * 64];
                }
            if (round == 1) {
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int j = 0; j < 4; ++j) accb[a][j] += theirs[(v++) * 64];
            }
        }
    }
    if (wk == 0) {
        const size_t slab = (size_t)blockIdx.x;
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const size_t tbase = (slab * p.ntaps_total + t) * p.F;
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int b = 0; b < SF; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = fb0 + (wf * 2 + a) * 16 + 4 * g + j;
                        const int sc = sb0 + b * 16 + lr;
                        p.partial[(tbase + f) * p.S + sc] = acc[t][a][b][j];
                    }
        }
        if (do_bias && lr == 0) {
#pragma unroll
            for (int a = 0; a < 2; ++a)
#pragma unroll
                for (int j = 0; j < 4; ++j) p.partial_bias[slab * p.F + fb0 + (wf * 2 + a) * 16 + 4 * g + j] = accb[a][j];
        }
    }
}


// ---- thin unit-stride layers: 32 x 32 channel blocks (FF = 2; 16 x 16 compiles, measured slower than wgrad2.hip) --------------------------------------------------------
// These layers wait for memory: a tile's MFMA work is a fraction of the time its bytes take at a CU's share of the HBM
// rate, so what matters is bytes in flight per CU and nothing else standing in the way.  Stages are half tiles as
// above (10 or 20 KB), in a ring of R slots with R - 1 stages in flight and counted waits, two workgroups per CU.
// The four waves split the TAPS (3 + 2 + 2 + 2), not the pixels: every wave reads all fixed-operand fragments of the
// stage and the shifted fragments of its own taps, holds at most 3 x FF x FF accumulators, and writes its taps'
// part of the slab itself - no combination pass through LDS at the end.  Rows are 32 FF bytes; 64-byte rows have
// their two 32-byte granules swapped for columns with bit 2 set (pixels four apart otherwise share banks in a
// transposed read of eight consecutive pixels), 32-byte rows need nothing.
template <int FF, int R>
__global__ __launch_bounds__(256, 2) void wgrad_thin_kernel(const WgradParams p) {
    constexpr int RB = 32 * FF;                      // bytes per LDS row (16 FF bf16 channels), both operands
    constexpr int PD = 18, SPX = 10 * PD;
    constexpr int LPR = RB / 16, PPI = 64 / LPR;     // lanes per row, pixels per DMA wave-instruction
    constexpr int NIF = 128 / PPI, NIS = (SPX + PPI - 1) / PPI;
    constexpr int FBYTES = NIF * 1024, SBYTES = NIS * 1024, SLOT = FBYTES + SBYTES;
    constexpr int NUF = NIF / 4, NUS = (NIS + 3) / 4;
    constexpr int PT = NUF + NUS;                    // loads per wave and stage (waves past the end repeat an instruction)
    constexpr int LA = R - 1;
    extern __shared__ __attribute__((aligned(1024))) char smem[];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int lr = lane & 15, g = lane >> 4;
    const int fb0 = blockIdx.y * 16 * FF, sb0 = blockIdx.z * 16 * FF;
    const int H = p.bd[1], W = p.bd[2];
    const int tap0 = wave == 0 ? 0 : 1 + 2 * wave, ntap = wave == 0 ? 3 : 2;   // taps 0-2, 3-4, 5-6, 7-8

    v4f acc[3][FF][FF];
    v4f accb[FF];
#pragma unroll
    for (int t = 0; t < 3; ++t)
#pragma unroll
        for (int a = 0; a < FF; ++a)
#pragma unroll
            for (int b = 0; b < FF; ++b) acc[t][a][b] = v4f{0.f, 0.f, 0.f, 0.f};
#pragma unroll
    for (int a = 0; a < FF; ++a) accb[a] = v4f{0.f, 0.f, 0.f, 0.f};
    const bool do_bias = wave == 3 && blockIdx.z == 0;

    const int fpb = p.f0_pitch * 2, spb = p.s_pitch * 2;
    const char* const fbase = p.f0 + (size_t)(p.f0_coff + fb0) * 2;
    const char* const sbase = p.s + (size_t)(p.s_coff + sb0) * 2;
    const char* const zero = reinterpret_cast<const char*>(g_zero_page_t) + (lane & 7) * 16;

    // ---- per-lane DMA constants ------------------------------------------------------------------------------------
    const int sl = lane % LPR;
    auto src_chunk = [&](int x) { return FF == 2 ? ((((sl >> 1) ^ gswz64(x)) << 1) | (sl & 1)) : sl; };
    int f_off[NUF], f_rc[NUF];
#pragma unroll
    for (int u = 0; u < NUF; ++u) {
        const int ps = (wave + 4 * u) * PPI + lane / LPR;
        const int r = ps >> 4, c = ps & 15;
        f_off[u] = (r * W + c) * fpb + src_chunk(c) * 16;
        f_rc[u] = (r << 8) | c;
    }
    int s_off[NUS], s_yx[NUS], s_dst[NUS];
#pragma unroll
    for (int u = 0; u < NUS; ++u) {
        const int i = (wave + 4 * u) % NIS;
        const int ps = i * PPI + lane / LPR;
        s_dst[u] = FBYTES + i * 1024;
        s_off[u] = 0; s_yx[u] = -1;
        if (ps < SPX) {
            const int py = ps / PD, px = ps - py * PD;
            s_off[u] = (py * W + px) * spb + src_chunk(px) * 16;
            s_yx[u] = (py << 8) | px;
        }
    }

    const int tiles_per_img = p.ntile[1] * p.ntile[2];
    auto magic = [](int d) { return d <= 1 ? 0u : (unsigned)(0x100000000ull / (unsigned)d) + 1u; };
    auto fdiv = [](int n, unsigned mg) { return mg ? (int)__umulhi((unsigned)n, mg) : n; };
    const unsigned mg_img = magic(tiles_per_img), mg_x = magic(p.ntile[2]);

    // stage j of this workgroup: half (j & 1) of tile blockIdx.x + (j >> 1) * nsplit
    auto issue_stage = [&](int j, int slot) {
        const int tile = blockIdx.x + (j >> 1) * p.nsplit;
        const int img = fdiv(tile, mg_img);
        const int tr = tile - img * tiles_per_img;
        const int tyi = fdiv(tr, mg_x);
        const int txi = tr - tyi * p.ntile[2];
        const int by0 = (tyi << 4) + 8 * (j & 1), bx0 = txi << 4;
        char* dst = smem + slot * SLOT;
        {
            const char* fb = fbase + (((long long)img * H + by0) * W + bx0) * fpb;
#pragma unroll
            for (int u = 0; u < NUF; ++u) {
                const bool ok = by0 + (f_rc[u] >> 8) < H && bx0 + (f_rc[u] & 255) < W;
                dma16t(ok ? fb + f_off[u] : zero, dst + (wave + 4 * u) * 1024);
            }
        }
        {
            const int iy0 = by0 - 1, ix0 = bx0 - 1;
            const char* sb = sbase + (((long long)img * H + iy0) * W + ix0) * spb;
#pragma unroll
            for (int u = 0; u < NUS; ++u) {
                const bool ok = s_yx[u] >= 0 && (unsigned)(iy0 + (s_yx[u] >> 8)) < (unsigned)H && (unsigned)(ix0 + (s_yx[u] & 255)) < (unsigned)W;
                dma16t(ok ? sb + s_off[u] : zero, dst + s_dst[u]);
            }
        }
    };

    // ---- per-lane fragment addresses: contraction slot 8 g + 4 h + q takes pixel v = 16 (g / 2) + 8 h + 4 (g % 2) + q -------
    const int q = lr >> 2, pp = lr & 3;
    int fa[2][FF], sa[2][3][FF];
#pragma unroll
    for (int h = 0; h < 2; ++h) {
        const int v = 16 * (g >> 1) + 8 * h + 4 * (g & 1) + q;
        const int r = v >> 4, c = v & 15;
#pragma unroll
        for (int a = 0; a < FF; ++a) fa[h][a] = v * RB + (FF == 2 ? (a ^ gswz64(c)) * 32 : 0) + pp * 8;
#pragma unroll
        for (int i = 0; i < 3; ++i) {
            const int t = tap0 + (i < ntap ? i : 0);
            const int dy = t / 3, dx = t - 3 * dy;
#pragma unroll
            for (int b = 0; b < FF; ++b)
                sa[h][i][b] = ((r + dy) * PD + c + dx) * RB + (FF == 2 ? (b ^ gswz64(c + dx)) * 32 : 0) + pp * 8 + FBYTES;
        }
    }

    const int ntw = p.tiles_total > (int)blockIdx.x ? (p.tiles_total - 1 - (int)blockIdx.x) / p.nsplit + 1 : 0;
    const int nstage = 2 * ntw;
    for (int j = 0; j < LA && j < nstage; ++j) issue_stage(j, j);
    int slot = 0, slot_issue = LA % R;
    for (int j = 0; j < nstage; ++j) {
        const int ahead = min(LA - 1, nstage - 1 - j);
        if (ahead >= 2) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(2 * PT) : "memory");
        else if (ahead == 1) asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PT) : "memory");
        else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();                // stage j has landed everywhere, stage j - 1 has been consumed
        asm volatile("" ::: "memory");
        if (j + LA < nstage) issue_stage(j + LA, slot_issue);
        slot_issue = slot_issue + 1 == R ? 0 : slot_issue + 1;
        const char* sm = smem + slot * SLOT;
        slot = slot + 1 == R ? 0 : slot + 1;
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) {
            uint4 afr[FF];
#pragma unroll
            for (int a = 0; a < FF; ++a) {
                v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(sm + fa[0][a] + ks * 32 * RB));
                v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(sm + fa[1][a] + ks * 32 * RB));
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
            for (int i = 0; i < 3; ++i) {
                if (i < ntap) {
                    uint4 bfr[FF];
#pragma unroll
                    for (int b = 0; b < FF; ++b) {
                        v4s lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(sm + sa[0][i][b] + ks * 2 * PD * RB));
                        v4s hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) v4s*)(sm + sa[1][i][b] + ks * 2 * PD * RB));
                        v8s f = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                        bfr[b] = __builtin_bit_cast(uint4, f);
                    }
#pragma unroll
                    for (int a = 0; a < FF; ++a)
#pragma unroll
                        for (int b = 0; b < FF; ++b)
                            acc[i][a][b] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(v8bf, afr[a]), __builtin_bit_cast(v8bf, bfr[b]), acc[i][a][b], 0, 0, 0);
                }
            }
        }
    }

    // ---- every wave writes its taps of the workgroup's slab ----------------------------------------------------------
    const size_t slab = (size_t)blockIdx.x;
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        if (i < ntap) {
            const size_t tbase = (slab * p.ntaps_total + tap0 + i) * p.F;
#pragma unroll
            for (int a = 0; a < FF; ++a)
#pragma unroll
                for (int b = 0; b < FF; ++b)
#pragma unroll
                    for (int j = 0; j < 4; ++j) {
                        const int f = fb0 + a * 16 + 4 * g + j;
                        const int sx = sb0 + b * 16 + lr;
                        p.partial[(tbase + f) * p.S + sx] = acc[i][a][b][j];
                    }
        }
    }
    if (do_bias && lr == 0) {
#pragma unroll
        for (int a = 0; a < FF; ++a)
#pragma unroll
            for (int j = 0; j < 4; ++j) p.partial_bias[slab * p.F + fb0 + a * 16 + 4 * g + j] = accb[a][j];
    }
}

}  // namespace

// Which kernel of this file serves `p`: 0 none, 1 unit stride (64 x 64 blocks), 2 / 3 gather stride 2 with 64 x 32 /
// 64 x 16 blocks.  Every shape condition the kernels rely on is checked here and nowhere else.
static int thick_kind(int dtype, const WgradParams& p) {
    static const bool on = getenv("MI_NO_WGRAD_THICK") == nullptr && !(getenv("MI_NO_WGRAD_DMA") && atoi(getenv("MI_NO_WGRAD_DMA")) != 0);
    static const bool s2on = getenv("MI_NO_WGRAD_THICK_S2") == nullptr;
    if (!on || !dtype || p.ntaps_total != 9 || p.ntapg != 1) return 0;
    if (p.bd[0] != 1 || p.tlog[0] != 0 || p.tlog[1] != 4 || p.tlog[2] != 4 || p.omin[1] != -1 || p.omin[2] != -1) return 0;
    if (p.F % 64 || (p.f0_c < p.F && p.f0_c % 8)) return 0;
    const long long fpm = p.f0_c < p.F && p.f1_pitch > p.f0_pitch ? p.f1_pitch : p.f0_pitch;
    if (p.ss == 1) {
        if (p.pd[1] != 18 || p.pd[2] != 18 || p.S % 64) return 0;
        for (int t = 0; t < 9; ++t)
            if (p.tap_off[t] != (t / 3) * 18 + t % 3) return 0;
        if (p.sd[1] != p.bd[1] || p.sd[2] != p.bd[2] || p.bd[1] > 255 * 16 || p.bd[2] > 255 * 16) return 0;
        if ((long long)p.bd[1] * p.bd[2] * (p.s_pitch > fpm ? p.s_pitch : fpm) * 2 >= (1ll << 30)) return 0;
        return 1;
    }
    if (p.ss == 2 && s2on) {
        if (p.pd[1] != 33 || p.pd[2] != 33 || p.S % 16) return 0;
        for (int t = 0; t < 9; ++t)
            if (p.tap_off[t] != (t / 3) * 33 + t % 3) return 0;
        if (p.bd[1] > 127 * 16 || p.bd[2] > 127 * 16) return 0;
        if ((long long)p.sd[1] * p.sd[2] * p.s_pitch * 2 >= (1ll << 30) || (long long)p.bd[1] * p.bd[2] * fpm * 2 >= (1ll << 30)) return 0;
        return p.S % 32 == 0 ? 2 : 3;
    }
    return 0;
}

// Block shape (fragments per wave, waves per side) under which the kernels of this file serve `p`, or false.  The
// split policy and the slab count follow the block shape, so wgrad_config asks here first.
bool wgrad_thick_config(int dtype, const WgradParams& p, int* ff, int* sf, int* wf, int* wsn) {
    const int k = thick_kind(dtype, p);
    if (!k) return false;
    *ff = 2; *wf = 2;
    *sf = k == 3 ? 1 : 2;
    *wsn = k == 1 ? 2 : 1;
    return true;
}

// Thin unit-stride layers: block shape (2,2,1,1) of the general policy.
static bool thin_serves(int dtype, const WgradParams& p, int ff, int sf, int wf, int wsn) {
    static const bool on = getenv("MI_NO_WGRAD_THIN") == nullptr && getenv("MI_NO_WGRAD_THICK") == nullptr &&
                           !(getenv("MI_NO_WGRAD_DMA") && atoi(getenv("MI_NO_WGRAD_DMA")) != 0);
    // 16 x 16 blocks stay with the whole-tile ring of wgrad2.hip: a 10 KB half-tile stage is too little work per barrier
    if (!on || !dtype || wf != 1 || wsn != 1 || ff != 2 || sf != 2) return false;
    if (p.ntaps_total != 9 || p.ntapg != 1 || p.ss != 1) return false;
    if (p.bd[0] != 1 || p.tlog[0] != 0 || p.tlog[1] != 4 || p.tlog[2] != 4 || p.omin[1] != -1 || p.omin[2] != -1) return false;
    if (p.pd[1] != 18 || p.pd[2] != 18 || p.F % (16 * ff) || p.S % (16 * ff) || p.f0_c < p.F) return false;
    for (int t = 0; t < 9; ++t)
        if (p.tap_off[t] != (t / 3) * 18 + t % 3) return false;
    if (p.sd[1] != p.bd[1] || p.sd[2] != p.bd[2] || p.bd[1] > 255 * 16 || p.bd[2] > 255 * 16) return false;
    if ((long long)p.bd[1] * p.bd[2] * (p.s_pitch > p.f0_pitch ? p.s_pitch : p.f0_pitch) * 2 >= (1ll << 30)) return false;
    return true;
}

// step-table name of the kernel of this file that serves p under block shape (ff, sf, wf, wsn); buf is left alone when none does
void wgrad_thick_label(int dtype, const WgradParams& p, int ff, int sf, int wf, int wsn, char* buf, size_t n) {
    if (thin_serves(dtype, p, ff, sf, wf, wsn)) { snprintf(buf, n, "wgrad_thin_kernel<2>"); return; }
    const int k = thick_kind(dtype, p);
    if (k == 1) snprintf(buf, n, "wgrad_thick_kernel");
    else if (k) snprintf(buf, n, "wgrad_thick_s2_kernel<%d>", k == 3 ? 1 : 2);
}

// 0 launched, 1 not served
static int launch_wgrad_thin(int dtype, const WgradParams& p, int ff, int sf, int wf, int wsn, hipStream_t st) {
    if (!thin_serves(dtype, p, ff, sf, wf, wsn)) return 1;
    const dim3 grid(p.nsplit, p.F / (16 * ff), p.S / (16 * ff));
    constexpr int R = 3;
    const size_t lds = (size_t)R * (8 + 12) * 1024;
    {   // per (kernel, device) inside mi_func_set_lds: a second GPU of the process gets the attribute too
        hipError_t e = mi_func_set_lds(reinterpret_cast<const void*>(wgrad_thin_kernel<2, R>), (int)lds);
        if (e != hipSuccess) { mi_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return -2; }
    }
    hipLaunchKernelGGL((wgrad_thin_kernel<2, R>), grid, dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("wgrad_thin launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}

// Returns 0 when launched, 1 when this file does not serve the shape (the caller falls back), < 0 on error.
int launch_wgrad_thick(int dtype, const WgradParams& p, int ff, int sf, int wf, int wsn, hipStream_t st) {
    {
        const int r = launch_wgrad_thin(dtype, p, ff, sf, wf, wsn, st);
        if (r <= 0) return r;
    }
    int cf, cs, cwf, cws;
    if (!wgrad_thick_config(dtype, p, &cf, &cs, &cwf, &cws)) return 1;
    if (cf != ff || cs != sf || cwf != wf || cws != wsn) { mi_set_error("wgrad_thick: block shape differs from the one the slabs were sized for"); return -2; }
    if (p.ss == 2) {
        const size_t lds2 = 2 * 16 * 1024 + 4 * (sf == 2 ? 10 : 5) * 1024;
        auto kern2 = sf == 2 ? wgrad_thick_s2_kernel<2> : wgrad_thick_s2_kernel<1>;
        {
            hipError_t e = mi_func_set_lds(reinterpret_cast<const void*>(kern2), (int)lds2);
            if (e != hipSuccess) { mi_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return -2; }
        }
        dim3 grid2(p.nsplit, p.F / 64, p.S / (16 * sf));
        hipLaunchKernelGGL(kern2, grid2, dim3(256), lds2, st, p);
        hipError_t e2 = hipGetLastError();
        if (e2 != hipSuccess) { mi_set_error(std::string("wgrad_thick_s2 launch: ") + hipGetErrorString(e2)); return -2; }
        return 0;
    }
    constexpr size_t lds = 2 * (16 + 23) * 1024;
    {
        hipError_t e = mi_func_set_lds(reinterpret_cast<const void*>(wgrad_thick_kernel), (int)lds);
        if (e != hipSuccess) { mi_set_error(std::string("hipFuncSetAttribute: ") + hipGetErrorString(e)); return -2; }
    }
    dim3 grid(p.nsplit, p.F / 64, p.S / 64);
    hipLaunchKernelGGL(wgrad_thick_kernel, grid, dim3(256), lds, st, p);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string("wgrad_thick launch: ") + hipGetErrorString(e)); return -2; }
    return 0;
}
