// Parameter blocks of the loss / metric / optimizer / sliding-window kernels.
#pragma once
#include <hip/hip_runtime.h>

// target dtype codes: 0 fp32, 1 uint8, 2 int64, 3 int32
struct LossParams {
    const float* logits;        // [nb][npix] fp32 (one channel)
    const void* target; int tdtype; int fold2;   // fold2: label value 2 counts as 1
    double* sums;               // [nb][4] = { sum p*t, sum p, sum t, sum bce } (fp64 accumulators)
    int fuse_finalize;          // the loss value is written by the backward launch (one-call training step) instead of a launch of its own
    int sums_cleared;           // forward: the caller has zeroed `sums` already (launch_loss_fwd then enqueues no memset)
    float* loss; float* loss_accum;
    float* dlogits; const float* gout; float gscale;
    int nb, npix;
    float smooth_nr, smooth_dr, lambda_dice, lambda_ce;
};

struct CountParams {
    const float* pred; const void* target; int tdtype; int fold2;
    int pred_is_binary; float thresh;   // binary: prediction == 1 ; else prediction >= thresh
    unsigned long long* counts;         // [nb][4] = { tp, fp, fn, tn }
    int nb, npix;
};

struct AdamParams {
    float* param; const float* grad; float* m; float* v;
    long long n;
    float decay;          // 1 - lr*weight_decay
    float beta1, beta2, eps;
    float step_size;      // lr / (1 - beta1^t)
    float inv_bc2_sqrt;   // 1 / sqrt(1 - beta2^t)
    float gscale;         // multiplies the gradient first (1/world_size under data parallelism)
};

struct SwParams {
    const float* image; float* windows;     // gather: image -> windows
    const float* win_out; float* out;       // blend: per-window predictions -> image
    const int* starts;                      // device [nwin][3] window starts in the padded image
    int nwin, win0, nb;
    int img[3], roi[3], pad_lo[3];
    float cval;
};

int launch_loss_fwd(const LossParams& p, hipStream_t st);
int launch_loss_bwd(const LossParams& p, hipStream_t st);
int launch_dice_counts(const CountParams& p, hipStream_t st);
int launch_adamw(const AdamParams& p, hipStream_t st);
int launch_sw_gather(const SwParams& p, int nwin_launch, hipStream_t st);
int launch_sw_blend(const SwParams& p, hipStream_t st);
int launch_add(int dtype, char* dst, int dpitch, int dcoff, const char* src, int spitch, int scoff, int c,
               long long npix, hipStream_t st);
int launch_fold_labels(void* t, int tdtype, long long n, hipStream_t st);
