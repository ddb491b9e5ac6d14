// Host-side geometry of the gather-GEMM launches: turns a logical convolution into
// tap tables, tile shapes and LDS patch extents.
#pragma once
#include <string.h>

#include "kparams.h"

struct ConvShape {
    int ndim;        // 2 or 3
    int transposed;  // 0: convolution (pad = (k-1)/2), 1: transposed convolution k3 s2 p1 output_padding 1
    int ksize;       // 3 or 1
    int stride;      // 1 or 2
    int cin, cout;
    int nb;
    int in_dims[3];  // (D,H,W) of the layer input; D = 1 in 2-D
};

enum { DIR_FWD = 0, DIR_DGRAD = 1 };

inline int kext(const ConvShape& s, int d) { return (s.ndim == 2 && d == 0) ? 1 : s.ksize; }
inline int ktaps(const ConvShape& s) { return kext(s, 0) * kext(s, 1) * kext(s, 2); }

inline void conv_out_dims(const ConvShape& s, int od[3]) {
    for (int d = 0; d < 3; ++d) {
        if (s.ndim == 2 && d == 0) { od[d] = 1; continue; }
        if (s.transposed) od[d] = s.in_dims[d] * 2;
        else od[d] = (s.in_dims[d] + 2 * ((s.ksize - 1) / 2) - s.ksize) / s.stride + 1;
    }
}

inline int ceil_log2(int v) { int l = 0; while ((1 << l) < v) ++l; return l; }

// 256 base pixels per tile with power-of-two extents: the shape that stages the fewest LDS patch
// pixels over the whole grid (tiles x patch volume), subject to the patch fitting `max_pp` pixels.
// `sa` is the gather stride, halo[d] the tap extent (omax - omin) per dim.
inline void choose_tile(int ndim, const int bd[3], int sa, const int halo[3], long max_pp, int tlog[3]) {
    double best = -1.0;
    int bl[3] = {0, 4, 4};
    for (int lz = 0; lz <= (ndim == 3 ? 8 : 0); ++lz)
        for (int ly = 0; ly + lz <= 8; ++ly) {
            const int lx = 8 - lz - ly;
            const int l[3] = {lz, ly, lx};
            double tiles = 1.0, pp = 1.0;
            for (int d = 0; d < 3; ++d) {
                const int T = 1 << l[d];
                tiles *= (bd[d] + T - 1) / T;
                pp *= (double)(T - 1) * sa + halo[d] + 1;
            }
            if (pp > (double)max_pp) continue;
            const double cost = tiles * pp - 1e-3 * lx;  // ties: the widest x extent (coalescing)
            if (best < 0 || cost < best) { best = cost; bl[0] = lz; bl[1] = ly; bl[2] = lx; }
        }
    tlog[0] = bl[0]; tlog[1] = bl[1]; tlog[2] = bl[2];
}

struct TapList {
    int n;
    int off[MI_MAX_TAPS][3];
    int kidx[MI_MAX_TAPS];
    int nphase;
    int phase_off[MI_MAX_PHASE][3];
    int phase_start[MI_MAX_PHASE + 1];
};

// gather mode: one phase, taps in kernel order, offset = sign * (k - pad)
inline void taps_gather(const ConvShape& s, int sign, TapList& t) {
    memset(&t, 0, sizeof(t));
    t.nphase = 1;
    int n = 0;
    for (int kz = 0; kz < kext(s, 0); ++kz)
        for (int ky = 0; ky < kext(s, 1); ++ky)
            for (int kx = 0; kx < kext(s, 2); ++kx) {
                const int k[3] = {kz, ky, kx};
                for (int d = 0; d < 3; ++d) t.off[n][d] = sign * (k[d] - (kext(s, d) - 1) / 2);
                t.kidx[n] = n;
                ++n;
            }
    t.n = n;
    t.phase_start[0] = 0; t.phase_start[1] = n;
}

// phase mode (stride-2 scatter): per dim, even outputs use k=1 at offset 0; odd outputs use
// k=0 at offset +1 and k=2 at offset 0
inline void taps_phase(const ConvShape& s, TapList& t) {
    memset(&t, 0, sizeof(t));
    const int nd = s.ndim;
    t.nphase = 1 << nd;
    int n = 0;
    for (int ph = 0; ph < t.nphase; ++ph) {
        int a[3] = {0, 0, 0};
        if (nd == 3) { a[0] = (ph >> 2) & 1; a[1] = (ph >> 1) & 1; a[2] = ph & 1; }
        else { a[1] = (ph >> 1) & 1; a[2] = ph & 1; }
        for (int d = 0; d < 3; ++d) t.phase_off[ph][d] = a[d];
        t.phase_start[ph] = n;
        int cnt[3], kk[3][2], oo[3][2];
        for (int d = 0; d < 3; ++d) {
            if (nd == 2 && d == 0) { cnt[d] = 1; kk[d][0] = 0; oo[d][0] = 0; continue; }
            if (a[d] == 0) { cnt[d] = 1; kk[d][0] = 1; oo[d][0] = 0; }
            else { cnt[d] = 2; kk[d][0] = 0; oo[d][0] = 1; kk[d][1] = 2; oo[d][1] = 0; }
        }
        for (int iz = 0; iz < cnt[0]; ++iz)
            for (int iy = 0; iy < cnt[1]; ++iy)
                for (int ix = 0; ix < cnt[2]; ++ix) {
                    t.off[n][0] = oo[0][iz]; t.off[n][1] = oo[1][iy]; t.off[n][2] = oo[2][ix];
                    t.kidx[n] = (kk[0][iz] * kext(s, 1) + kk[1][iy]) * kext(s, 2) + kk[2][ix];
                    ++n;
                }
    }
    t.phase_start[t.nphase] = n;
    t.n = n;
}

// Fills the geometric part of an igemm launch and returns the source kernel index of every
// packed tap slot in kidx.  Channel / pointer fields are left to the caller.
inline bool igemm_geometry(const ConvShape& s, int dir, IgemmParams& p, int kidx[MI_MAX_TAPS], int* ntaps) {
    int od[3];
    conv_out_dims(s, od);
    TapList t;
    bool phase;
    if (!s.transposed) {
        if (dir == DIR_FWD) {
            taps_gather(s, +1, t); phase = false;
            for (int d = 0; d < 3; ++d) { p.id[d] = s.in_dims[d]; p.od[d] = od[d]; p.bd[d] = od[d]; }
            p.sa = s.stride; p.so = 1;
        } else if (s.stride == 1) {
            taps_gather(s, -1, t); phase = false;
            for (int d = 0; d < 3; ++d) { p.id[d] = od[d]; p.od[d] = s.in_dims[d]; p.bd[d] = s.in_dims[d]; }
            p.sa = 1; p.so = 1;
        } else {
            taps_phase(s, t); phase = true;
            for (int d = 0; d < 3; ++d) {
                p.id[d] = od[d]; p.od[d] = s.in_dims[d]; p.bd[d] = od[d];
                if (!(s.ndim == 2 && d == 0) && s.in_dims[d] != 2 * od[d]) return false;  // even sizes only
            }
            p.sa = 1; p.so = 2;
        }
    } else {
        if (dir == DIR_FWD) {
            taps_phase(s, t); phase = true;
            for (int d = 0; d < 3; ++d) { p.id[d] = s.in_dims[d]; p.od[d] = od[d]; p.bd[d] = s.in_dims[d]; }
            p.sa = 1; p.so = 2;
        } else {
            ConvShape g = s; g.transposed = 0;  // gather over dy with stride 2, offsets k-1
            taps_gather(g, +1, t); phase = false;
            for (int d = 0; d < 3; ++d) { p.id[d] = od[d]; p.od[d] = s.in_dims[d]; p.bd[d] = s.in_dims[d]; }
            p.sa = 2; p.so = 1;
        }
    }
    (void)phase;
    p.nb = s.nb;
    int omax[3], halo[3];
    for (int d = 0; d < 3; ++d) { p.omin[d] = 0; omax[d] = 0; }
    for (int i = 0; i < t.n; ++i)
        for (int d = 0; d < 3; ++d) {
            if (t.off[i][d] < p.omin[d]) p.omin[d] = t.off[i][d];
            if (t.off[i][d] > omax[d]) omax[d] = t.off[i][d];
        }
    for (int d = 0; d < 3; ++d) halo[d] = omax[d] - p.omin[d];
    choose_tile(s.ndim, p.bd, p.sa, halo, 2900, p.tlog);
    // 2-D one-tap layers: every tile shape stages the same pixels; the square tile is the one the asynchronously
    // staged kernels serve
    if (s.ndim == 2 && t.n == 1 && p.bd[1] >= 16 && p.bd[2] >= 16) { p.tlog[0] = 0; p.tlog[1] = 4; p.tlog[2] = 4; }
    for (int d = 0; d < 3; ++d) {
        const int T = 1 << p.tlog[d];
        p.ntile[d] = (p.bd[d] + T - 1) / T;
        p.pd[d] = (T - 1) * p.sa + (omax[d] - p.omin[d]) + 1;
    }
    p.nphase = t.nphase;
    int maxpt = 0;
    for (int ph = 0; ph < t.nphase; ++ph) {
        for (int d = 0; d < 3; ++d) p.phase_off[ph][d] = t.phase_off[ph][d];
        p.phase_start[ph] = t.phase_start[ph];
        const int c = t.phase_start[ph + 1] - t.phase_start[ph];
        if (c > maxpt) maxpt = c;
    }
    p.phase_start[t.nphase] = t.n;
    for (int i = 0; i < t.n; ++i) {
        p.tap_off[i] = ((t.off[i][0] - p.omin[0]) * p.pd[1] + (t.off[i][1] - p.omin[1])) * p.pd[2] +
                       (t.off[i][2] - p.omin[2]);
        kidx[i] = t.kidx[i];
    }
    p.comp_mode = 0;
    if (t.nphase == 1 && p.sa == 2 && s.ndim == 3 && s.ksize == 3) {
        // slots by polyphase component (see IgemmParams::comp_mode); both gathers with stride 2 (the stride-2
        // convolution and the data gradient of the transposed convolution) read input pixel 2 o + k - 1
        p.comp_mode = 1;
        for (int d = 0; d < 3; ++d) p.cpd[d] = (1 << p.tlog[d]) + 1;
        int n = 0;
        for (int c = 0; c < 8; ++c) {
            const int r[3] = {(c >> 2) & 1, (c >> 1) & 1, c & 1};
            p.comp_start[c] = n;
            int cnt[3], kk[3][2], oo[3][2];
            for (int d = 0; d < 3; ++d) {
                if (r[d] == 0) { cnt[d] = 1; kk[d][0] = 1; oo[d][0] = 0; }
                else { cnt[d] = 2; kk[d][0] = 0; oo[d][0] = 0; kk[d][1] = 2; oo[d][1] = 1; }
            }
            for (int iz = 0; iz < cnt[0]; ++iz)
                for (int iy = 0; iy < cnt[1]; ++iy)
                    for (int ix = 0; ix < cnt[2]; ++ix) {
                        const int k[3] = {kk[0][iz], kk[1][iy], kk[2][ix]};
                        kidx[n] = (k[0] * 3 + k[1]) * 3 + k[2];
                        p.tap_off[n] = (k[0] * p.pd[1] + k[1]) * p.pd[2] + k[2];       // omin = -1: patch offset = k
                        p.ctap_off[n] = (oo[0][iz] * p.cpd[1] + oo[1][iy]) * p.cpd[2] + oo[2][ix];
                        ++n;
                    }
        }
        p.comp_start[8] = n;
    } else
    // one-phase gathers: taps in ascending patch order (a data gradient mirrors the kernel, so its taps come out
    // reversed); kernels with compile-time tap offsets rely on it, the weight packing follows kidx
    if (t.nphase == 1)
        for (int i = 1; i < t.n; ++i)
            for (int j = i; j > 0 && p.tap_off[j - 1] > p.tap_off[j]; --j) {
                const int a = p.tap_off[j]; p.tap_off[j] = p.tap_off[j - 1]; p.tap_off[j - 1] = a;
                const int b = kidx[j]; kidx[j] = kidx[j - 1]; kidx[j - 1] = b;
            }
    p.tapg = maxpt < 9 ? maxpt : 9;
    *ntaps = t.n;
    return true;
}

// Geometric part of a weight-gradient launch.  Convolution: F = dy, S = x (stride = conv
// stride).  Transposed convolution: F = x, S = dy (stride 2).
inline void wgrad_geometry(const ConvShape& s, int esz, WgradParams& p) {
    int od[3];
    conv_out_dims(s, od);
    for (int d = 0; d < 3; ++d) {
        if (!s.transposed) { p.bd[d] = od[d]; p.sd[d] = s.in_dims[d]; }
        else { p.bd[d] = s.in_dims[d]; p.sd[d] = od[d]; }
    }
    p.ss = s.transposed ? 2 : s.stride;
    p.nb = s.nb;
    const int kz = kext(s, 0), ky = kext(s, 1), kx = kext(s, 2);
    {
        const int halo[3] = {0, ky - 1, kx - 1};  // one kz plane per tap group: no z halo
        const long row = 16 * esz + 16;           // smallest LDS row (16 channels)
        choose_tile(s.ndim, p.bd, p.ss, halo, (150 * 1024 - 256 * row) / row, p.tlog);
    }
    const int pz = (kz - 1) / 2, py = (ky - 1) / 2, px = (kx - 1) / 2;
    p.ntapg = kz;
    for (int g = 0; g < 3; ++g) p.tg_zoff[g] = (g < kz) ? g - pz : 0;
    p.omin[0] = 0; p.omin[1] = -py; p.omin[2] = -px;
    for (int d = 0; d < 3; ++d) {
        const int T = 1 << p.tlog[d];
        p.ntile[d] = (p.bd[d] + T - 1) / T;
    }
    p.pd[0] = ((1 << p.tlog[0]) - 1) * p.ss + 1;
    p.pd[1] = ((1 << p.tlog[1]) - 1) * p.ss + (ky - 1) + 1;
    p.pd[2] = ((1 << p.tlog[2]) - 1) * p.ss + (kx - 1) + 1;
    int n = 0;
    for (int iy = 0; iy < ky; ++iy)
        for (int ix = 0; ix < kx; ++ix) p.tap_off[n++] = iy * p.pd[2] + ix;
    p.ntaps_total = kz * ky * kx;
    p.tiles_total = p.nb * p.ntile[0] * p.ntile[1] * p.ntile[2];
}
