// Detection-metric matching on the device: for every image, the ground-truth boxes claim predictions one after the
// other (in their stored order); a ground truth takes the not-yet-claimed prediction of its own class with the largest
// IoU (first one on ties) and that prediction counts as a true positive when the IoU reaches the threshold.
// Arithmetic is the reference's numpy float32 arithmetic, one rounding per operation (build with -ffp-contract=off):
// IoU with the "+1" pixel convention, ground truths scaled from normalised cx,cy,w,h.
#include "yamd_common.h"

#include <cmath>

using namespace yamd;

namespace {

constexpr int MT = 256;

struct Best {
    float iou;
    int idx;
};

// argmax order of numpy: NaN beats every number, ties keep the lower index
__device__ __forceinline__ bool better(float a, int ai, float b, int bi)
{
    const bool an = a != a, bn = b != b;
    if (an || bn) return an && (!bn || ai < bi);
    return a > b || (a == b && ai < bi);
}

__global__ __launch_bounds__(MT) void detect_match_kernel(const float *rows, const int *counts, int max_out, int width,
                                                          const float *targets, const int *target_counts, int max_targets,
                                                          float img_w, float img_h, float iou_thr, unsigned char *tp)
{
    __shared__ float s_iou[MT / 64];
    __shared__ int s_idx[MT / 64];
    __shared__ int s_win;
    const long long b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int K = min(counts[b], max_out);
    const int G = target_counts ? min(target_counts[b], max_targets) : max_targets;
    const float *R = rows + b * (long long)max_out * width;
    unsigned char *T = tp + b * (long long)max_out;
    for (int i = tid; i < K; i += MT) T[i] = 0;
    __syncthreads();
    for (int g = 0; g < G; ++g) {
        const float *t = targets + (b * max_targets + g) * 5;
        const float tx = t[0], ty = t[1], tw = t[2], th = t[3], tc = t[4];
        float sum = tx + ty;
        sum = sum + tw;
        sum = sum + th;
        sum = sum + tc;
        if (!(sum != 0.0f)) continue;                       // all-zero rows are padding
        const float cx = tx * img_w, cy = ty * img_h, w = tw * img_w, h = th * img_h;
        const float hw = 0.5f * w, hh = 0.5f * h;
        const float gx1 = cx - hw, gy1 = cy - hh, gx2 = cx + hw, gy2 = cy + hh;
        const float ga = (gx2 - gx1 + 1.0f) * (gy2 - gy1 + 1.0f);
        float bi = 0.0f;
        int bidx = -1;
        for (int i = tid; i < K; i += MT) {
            const float *r = R + (long long)i * width;
            if (r[width - 1] != tc || T[i]) continue;
            const float x1 = r[0], y1 = r[1], x2 = r[2], y2 = r[3];
            const float lx = fmaxf(x1, gx1), ly = fmaxf(y1, gy1), rx = fminf(x2, gx2), ry = fminf(y2, gy2);
            const float iw = fmaxf(rx - lx + 1.0f, 0.0f), ih = fmaxf(ry - ly + 1.0f, 0.0f);
            const float inter = iw * ih;
            const float pa = (x2 - x1 + 1.0f) * (y2 - y1 + 1.0f);
            float uni = pa + ga;
            uni = uni - inter;
            const float iou = inter / uni;
            if (bidx < 0 || better(iou, i, bi, bidx)) { bi = iou; bidx = i; }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            const float oi = __shfl_xor(bi, d);
            const int ox = __shfl_xor(bidx, d);
            if (ox >= 0 && (bidx < 0 || better(oi, ox, bi, bidx))) { bi = oi; bidx = ox; }
        }
        if (lane == 0) { s_iou[wave] = bi; s_idx[wave] = bidx; }
        __syncthreads();
        if (tid == 0) {
            float fi = s_iou[0];
            int fx = s_idx[0];
            for (int wv = 1; wv < MT / 64; ++wv)
                if (s_idx[wv] >= 0 && (fx < 0 || better(s_iou[wv], s_idx[wv], fi, fx))) { fi = s_iou[wv]; fx = s_idx[wv]; }
            s_win = (fx >= 0 && fi >= iou_thr) ? fx : -1;
        }
        __syncthreads();
        if (tid == 0 && s_win >= 0) T[s_win] = 1;
        __syncthreads();
    }
}

}  // namespace

extern "C" int yamd_detect_match(const float *rows, const int *counts, int batch, int max_out, int width, const float *targets,
                                 const int *target_counts, int max_targets, float img_w, float img_h, float iou_thr,
                                 uint8_t *true_positive, void *stream)
{
    YAMD_REQUIRE(rows && counts && true_positive, "null argument");
    YAMD_REQUIRE(batch > 0 && max_out > 0 && width >= 7 && max_targets >= 0, "bad shape B=%d max_out=%d width=%d targets=%d", batch,
                 max_out, width, max_targets);
    YAMD_REQUIRE(max_targets == 0 || targets, "targets missing");
    hipLaunchKernelGGL(detect_match_kernel, dim3(batch), dim3(MT), 0, (hipStream_t)stream, rows, counts, max_out, width, targets,
                       target_counts, max_targets, img_w, img_h, iou_thr, true_positive);
    YAMD_LAUNCH_CHECK();
    return YAMD_OK;
}
