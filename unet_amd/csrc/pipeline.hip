// GPU-resident input pipeline for the 2-D training crops: intensity window, foreground bounding box,
// pad, positive / negative crop-centre selection, crop and 90-degree rotation.  Slices are planar
// [S][H][W] (fp32 or int16 images; fp32 / uint8 / int64 / int32 labels), outputs are fp32 [B][1][ch][cw].
#include "common.h"
#include "../../include/unet_mi355.h"

namespace {

__device__ __forceinline__ float load_img(const void* p, int dt, size_t i) {
    return dt == 1 ? (float)reinterpret_cast<const short*>(p)[i] : reinterpret_cast<const float*>(p)[i];
}
__device__ __forceinline__ float load_lbl(const void* t, int dt, size_t i) {
    switch (dt) {
        case 1: return (float)reinterpret_cast<const unsigned char*>(t)[i];
        case 2: return (float)reinterpret_cast<const long long*>(t)[i];
        case 3: return (float)reinterpret_cast<const int*>(t)[i];
        default: return reinterpret_cast<const float*>(t)[i];
    }
}

// bbox[s] = {y0, y1, x0, x1} (half open) of the pixels with img > thr; the whole slice when none is
__global__ __launch_bounds__(256) void fg_bbox_kernel(const void* img, int dt, float thr, int H, int W, int* bbox) {
    __shared__ int red[4][4];
    const int s = blockIdx.x;
    const size_t base = (size_t)s * H * W;
    int y0 = H, y1 = -1, x0 = W, x1 = -1;
    for (int i = threadIdx.x; i < H * W; i += 256) {
        if (load_img(img, dt, base + i) > thr) {
            const int y = i / W, x = i - y * W;
            y0 = min(y0, y); y1 = max(y1, y); x0 = min(x0, x); x1 = max(x1, x);
        }
    }
    for (int o = 32; o > 0; o >>= 1) {
        y0 = min(y0, __shfl_xor(y0, o, 64)); y1 = max(y1, __shfl_xor(y1, o, 64));
        x0 = min(x0, __shfl_xor(x0, o, 64)); x1 = max(x1, __shfl_xor(x1, o, 64));
    }
    const int w = threadIdx.x >> 6;
    if ((threadIdx.x & 63) == 0) { red[0][w] = y0; red[1][w] = y1; red[2][w] = x0; red[3][w] = x1; }
    __syncthreads();
    if (threadIdx.x == 0) {
        for (int k = 1; k < 4; ++k) {
            red[0][0] = min(red[0][0], red[0][k]); red[1][0] = max(red[1][0], red[1][k]);
            red[2][0] = min(red[2][0], red[2][k]); red[3][0] = max(red[3][0], red[3][k]);
        }
        if (red[1][0] < 0) { bbox[s * 4] = 0; bbox[s * 4 + 1] = H; bbox[s * 4 + 2] = 0; bbox[s * 4 + 3] = W; }
        else { bbox[s * 4] = red[0][0]; bbox[s * 4 + 1] = red[1][0] + 1; bbox[s * 4 + 2] = red[2][0]; bbox[s * 4 + 3] = red[3][0] + 1; }
    }
}

// One workgroup per output sample: counts the foreground (label > 0) and background (label <= 0 and
// image > bg_thr) pixels of the slice's bounding box row by row, picks the class from u[0] and the
// floor(u[1] * n)-th pixel of that class in row-major order.  centre = (y, x) inside the bounding box.
__global__ __launch_bounds__(256) void pick_centers_kernel(const void* img, int idt, const void* lbl, int ldt, float bg_thr,
                                                           const int* bbox, int H, int W, const int* sample_slice,
                                                           const float* u, float pos_ratio, int* centers) {
    extern __shared__ int sm[];   // [2][rows] per-row counts, then prefix
    __shared__ int pick[3];       // class, row, rank inside the row
    const int b = blockIdx.x;
    const int s = sample_slice[b];
    const int y0 = bbox[s * 4], y1 = bbox[s * 4 + 1], x0 = bbox[s * 4 + 2], x1 = bbox[s * 4 + 3];
    const int rows = y1 - y0, cols = x1 - x0;
    const size_t base = (size_t)s * H * W;
    int* cfg = sm; int* cbg = sm + rows;
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = wave; r < rows; r += 4) {
        int nf = 0, nb = 0;
        for (int c = lane; c < cols; c += 64) {
            const size_t i = base + (size_t)(y0 + r) * W + x0 + c;
            const float l = load_lbl(lbl, ldt, i);
            if (l > 0.f) ++nf; else if (load_img(img, idt, i) > bg_thr) ++nb;
        }
        for (int o = 32; o > 0; o >>= 1) { nf += __shfl_xor(nf, o, 64); nb += __shfl_xor(nb, o, 64); }
        if (lane == 0) { cfg[r] = nf; cbg[r] = nb; }
    }
    __syncthreads();
    if (threadIdx.x == 0) {
        long long tf = 0, tb = 0;
        for (int r = 0; r < rows; ++r) { tf += cfg[r]; tb += cbg[r]; }
        int cls;   // 1: foreground
        if (tf == 0 && tb == 0) cls = -1;
        else if (tf == 0) cls = 0;
        else if (tb == 0) cls = 1;
        else cls = u[b * 2] < pos_ratio ? 1 : 0;
        pick[0] = cls; pick[1] = 0; pick[2] = 0;
        if (cls >= 0) {
            const long long n = cls ? tf : tb;
            long long k = (long long)((double)u[b * 2 + 1] * (double)n);
            if (k >= n) k = n - 1;
            const int* cnt = cls ? cfg : cbg;
            int r = 0;
            while (k >= cnt[r]) { k -= cnt[r]; ++r; }
            pick[1] = r; pick[2] = (int)k;
        }
    }
    __syncthreads();
    const int cls = pick[0];
    if (cls < 0) {   // nothing to choose from: the centre of the box
        if (threadIdx.x == 0) { centers[b * 2] = rows / 2; centers[b * 2 + 1] = cols / 2; }
        return;
    }
    if (wave == 0) {
        const int r = pick[1];
        int k = pick[2];
        for (int c0 = 0; c0 < cols; c0 += 64) {
            const int c = c0 + lane;
            bool hit = false;
            if (c < cols) {
                const size_t i = base + (size_t)(y0 + r) * W + x0 + c;
                const float l = load_lbl(lbl, ldt, i);
                hit = cls ? (l > 0.f) : (!(l > 0.f) && load_img(img, idt, i) > bg_thr);
            }
            const unsigned long long m = __ballot(hit);
            const int n = __popcll(m);
            if (k < n) {
                // the k-th set bit of the mask
                const int before = __popcll(m & ((1ull << lane) - 1ull));
                if (hit && before == k) { centers[b * 2] = r; centers[b * 2 + 1] = c; }
                return;
            }
            k -= n;
        }
    }
}

// out[b][y][x] for the rotated crop: desc[b] = {slice, oy, ox, k}: (oy, ox) is the crop origin in source
// pixel coordinates (may be negative or run past the bounding box: zero padding), k quarter turns
// counter-clockwise as numpy.rot90 does.  box[b] = bounding box of the slice (pixels outside it are padding).
__global__ __launch_bounds__(256) void crop_scale_rot_kernel(const void* img, int idt, const void* lbl, int ldt, int H, int W,
                                                             const int* desc, const int* bbox, float a_min, float a_range,
                                                             float b_min, float b_range, int clip, int ch, int cw,
                                                             float* out_img, float* out_lbl) {
    const int b = blockIdx.y;
    const int s = desc[b * 4], oy = desc[b * 4 + 1], ox = desc[b * 4 + 2], k = desc[b * 4 + 3] & 3;
    const int y0 = bbox[s * 4], y1 = bbox[s * 4 + 1], x0 = bbox[s * 4 + 2], x1 = bbox[s * 4 + 3];
    const size_t base = (size_t)s * H * W;
    for (int i = blockIdx.x * 256 + threadIdx.x; i < ch * cw; i += gridDim.x * 256) {
        const int y = i / cw, x = i - y * cw;
        // rot90 (counter-clockwise, axes (0,1)) of a ch x cw crop: out[y][x] = in[iy][ix]
        int iy, ix;
        if (k == 0) { iy = y; ix = x; }
        else if (k == 1) { iy = x; ix = cw - 1 - y; }
        else if (k == 2) { iy = ch - 1 - y; ix = cw - 1 - x; }
        else { iy = ch - 1 - x; ix = y; }
        const int sy = oy + iy, sx = ox + ix;
        float v = 0.f, l = 0.f;
        if (sy >= y0 && sy < y1 && sx >= x0 && sx < x1) {
            const size_t j = base + (size_t)sy * W + sx;
            v = __fdiv_rn(load_img(img, idt, j) - a_min, a_range);      // IEEE division, as (img - a_min) / (a_max - a_min)
            v = __fadd_rn(__fmul_rn(v, b_range), b_min);                // no fused multiply-add: matches separate ops
            if (clip) v = fminf(fmaxf(v, b_min), b_min + b_range);
            l = load_lbl(lbl, ldt, j);
        }
        out_img[(size_t)b * ch * cw + i] = v;
        out_lbl[(size_t)b * ch * cw + i] = l;
    }
}

int launched(const char* what) {
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) { mi_set_error(std::string(what) + ": " + hipGetErrorString(e)); return -2; }
    return 0;
}

}  // namespace

extern "C" {

int mi_fg_bbox(const void* img, int img_dtype, float thr, int ns, int H, int W, int32_t* bbox, void* stream) {
    MI_CHECK(img && bbox && ns > 0 && H > 0 && W > 0, "bad argument");
    MI_CHECK(img_dtype == 0 || img_dtype == 1, "image dtype must be 0 (fp32) or 1 (int16)");
    hipLaunchKernelGGL(fg_bbox_kernel, dim3(ns), dim3(256), 0, (hipStream_t)stream, img, img_dtype, thr, H, W, bbox);
    return launched("fg_bbox");
}

int mi_pick_centers(const void* img, int img_dtype, const void* lbl, int lbl_dtype, float bg_thr, const int32_t* bbox, int H,
                    int W, const int32_t* sample_slice, const float* u, float pos_ratio, int nsamp, int32_t* centers,
                    void* stream) {
    MI_CHECK(img && lbl && bbox && sample_slice && u && centers && nsamp > 0, "bad argument");
    MI_CHECK(img_dtype == 0 || img_dtype == 1, "image dtype must be 0 (fp32) or 1 (int16)");
    MI_CHECK(lbl_dtype >= 0 && lbl_dtype <= 3, "label dtype must be 0..3");
    MI_CHECK(H <= 8192, "slice height above 8192");
    hipLaunchKernelGGL(pick_centers_kernel, dim3(nsamp), dim3(256), (size_t)2 * H * sizeof(int), (hipStream_t)stream, img,
                       img_dtype, lbl, lbl_dtype, bg_thr, bbox, H, W, sample_slice, u, pos_ratio, centers);
    return launched("pick_centers");
}

int mi_crop_scale_rot(const void* img, int img_dtype, const void* lbl, int lbl_dtype, int H, int W, const int32_t* desc,
                      const int32_t* bbox, int nsamp, float a_min, float a_max, float b_min, float b_max, int clip, int ch,
                      int cw, float* out_img, float* out_lbl, void* stream) {
    MI_CHECK(img && lbl && desc && bbox && out_img && out_lbl && nsamp > 0 && ch > 0 && cw > 0, "bad argument");
    MI_CHECK(img_dtype == 0 || img_dtype == 1, "image dtype must be 0 (fp32) or 1 (int16)");
    MI_CHECK(lbl_dtype >= 0 && lbl_dtype <= 3, "label dtype must be 0..3");
    MI_CHECK(a_max > a_min, "a_max must exceed a_min");
    MI_CHECK(ch == cw, "rotation by quarter turns needs a square crop");
    int gx = (ch * cw + 1023) / 1024; if (gx < 1) gx = 1;
    hipLaunchKernelGGL(crop_scale_rot_kernel, dim3(gx, nsamp), dim3(256), 0, (hipStream_t)stream, img, img_dtype, lbl, lbl_dtype,
                       H, W, desc, bbox, a_min, a_max - a_min, b_min, b_max - b_min, clip, ch, cw, out_img, out_lbl);
    return launched("crop_scale_rot");
}

}  // extern "C"
