/*
 * TEST INFRASTRUCTURE ONLY -- CPU oracle for the detection post-processing path.
 *
 * This file is a plain-C restatement of the candidate filter + greedy NMS that the
 * reference runs after its heads.  It is used by tests/, by __graft_entry__.smoke()
 * and by bench.py's cpu_baseline leg as the checker / timed CPU baseline.  The
 * shipped library (yolo_amd/csrc) never links or calls anything in here.
 *
 * What it follows:
 *   - luoyang/utils/torch_utils.py:101-141   non_max_suppression
 *   - luoyang/utils/torch_utils.py:144-191   non_max_suppression_with_point
 *   - luoyang/utils/torch_utils.py:194-241   non_max_suppression_with_point_conf
 *   - torchvision.ops.nms / box_convert (third-party, NOT in /root/reference, no pinned
 *     version anywhere in the reference; "parity unpinned" for that dependency itself).
 *     The published CPU algorithm is restated in yo_nms(): stable descending sort of the
 *     scores, areas (x2-x1)*(y2-y1) without +1, greedy sweep, float IoU compared with the
 *     double threshold using strict '>', so 0/0 = NaN never suppresses; kept indices are
 *     returned in descending score order.
 *
 * All arithmetic is IEEE binary32 with one rounding per operation (build with
 * -ffp-contract=off so no fused multiply-add is introduced).
 */
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <math.h>

/* stable merge sort of idx[0..n) by descending score (ties keep ascending index) */
static void merge_sort_desc(const float *s, int64_t *idx, int64_t *tmp, int64_t n)
{
    for (int64_t w = 1; w < n; w *= 2) {
        for (int64_t lo = 0; lo < n; lo += 2 * w) {
            int64_t mid = lo + w < n ? lo + w : n;
            int64_t hi = lo + 2 * w < n ? lo + 2 * w : n;
            int64_t a = lo, b = mid, o = lo;
            while (a < mid && b < hi) {
                /* take from the right run only when strictly greater: stability */
                if (s[idx[b]] > s[idx[a]]) tmp[o++] = idx[b++];
                else tmp[o++] = idx[a++];
            }
            while (a < mid) tmp[o++] = idx[a++];
            while (b < hi) tmp[o++] = idx[b++];
        }
        memcpy(idx, tmp, (size_t)n * sizeof(int64_t));
    }
}

/* torchvision.ops.nms, CPU kernel semantics.  boxes: n x 4 (x1,y1,x2,y2).  Returns K and
 * writes K kept indices (descending score order) into keep. */
int64_t yo_nms(const float *boxes, const float *scores, int64_t n, double iou_thr, int64_t *keep)
{
    if (n <= 0) return 0;
    int64_t *order = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    int64_t *tmp = (int64_t *)malloc((size_t)n * sizeof(int64_t));
    float *area = (float *)malloc((size_t)n * sizeof(float));
    unsigned char *sup = (unsigned char *)calloc((size_t)n, 1);
    for (int64_t i = 0; i < n; ++i) {
        order[i] = i;
        float w = boxes[4 * i + 2] - boxes[4 * i + 0];
        float h = boxes[4 * i + 3] - boxes[4 * i + 1];
        area[i] = w * h;
    }
    merge_sort_desc(scores, order, tmp, n);
    int64_t k = 0;
    for (int64_t _i = 0; _i < n; ++_i) {
        int64_t i = order[_i];
        if (sup[i]) continue;
        keep[k++] = i;
        float ix1 = boxes[4 * i], iy1 = boxes[4 * i + 1], ix2 = boxes[4 * i + 2], iy2 = boxes[4 * i + 3];
        float ia = area[i];
        for (int64_t _j = _i + 1; _j < n; ++_j) {
            int64_t j = order[_j];
            if (sup[j]) continue;
            float xx1 = ix1 > boxes[4 * j] ? ix1 : boxes[4 * j];
            float yy1 = iy1 > boxes[4 * j + 1] ? iy1 : boxes[4 * j + 1];
            float xx2 = ix2 < boxes[4 * j + 2] ? ix2 : boxes[4 * j + 2];
            float yy2 = iy2 < boxes[4 * j + 3] ? iy2 : boxes[4 * j + 3];
            float w = xx2 - xx1; w = w > 0.0f ? w : 0.0f;
            float h = yy2 - yy1; h = h > 0.0f ? h : 0.0f;
            float inter = w * h;
            float uni = ia + area[j];
            uni = uni - inter;
            float ovr = inter / uni;
            if ((double)ovr > iou_thr) sup[j] = 1;
        }
    }
    free(order); free(tmp); free(area); free(sup);
    return k;
}

/*
 * One image of non_max_suppression{,_with_point,_with_point_conf}.
 *   pred      A x C row-major: [b0,b1,b2,b3,obj, extra..., cls...]
 *   n_extra   columns between obj and the class scores (0, 2P or 3P)
 *   in_cxcywh 1: boxes are (cx,cy,w,h) and converted as c -/+ 0.5*wh; 0: already xyxy
 *   rows      output, K x (6 + n_extra + 1): [x1,y1,x2,y2,obj,cls_conf, extra..., cls_id]
 *   keep_src  output, K source row indices into pred (may be NULL)
 * Returns K (0 means the reference returns None for this image).
 */
int64_t yo_filter_nms(const float *pred, int64_t A, int64_t C, float img_size, int in_cxcywh,
                      int64_t n_extra, float conf_thr, double nms_thr, float *rows, int64_t *keep_src)
{
    const int64_t cls0 = 5 + n_extra;
    const int64_t nc = C - cls0;
    const int64_t W = 6 + n_extra + 1;
    const float hi = img_size - 1.0f;
    float *det = (float *)malloc((size_t)A * (size_t)W * sizeof(float));
    float *obox = (float *)malloc((size_t)A * 4 * sizeof(float));
    float *score = (float *)malloc((size_t)A * sizeof(float));
    int64_t *src = (int64_t *)malloc((size_t)A * sizeof(int64_t));
    int64_t *keep = (int64_t *)malloc((size_t)A * sizeof(int64_t));
    int64_t n = 0;
    for (int64_t a = 0; a < A; ++a) {
        const float *p = pred + a * C;
        float b[4];
        if (in_cxcywh) {
            float hw = 0.5f * p[2], hh = 0.5f * p[3];
            b[0] = p[0] - hw; b[1] = p[1] - hh; b[2] = p[0] + hw; b[3] = p[1] + hh;
        } else {
            b[0] = p[0]; b[1] = p[1]; b[2] = p[2]; b[3] = p[3];
        }
        for (int q = 0; q < 4; ++q) {           /* torch.clamp(min=0, max=img-1) */
            float v = b[q];
            v = v < 0.0f ? 0.0f : v;
            v = v > hi ? hi : v;
            b[q] = v;
        }
        float best = p[cls0]; int64_t bi = 0;     /* torch.max: first maximal index */
        for (int64_t c = 1; c < nc; ++c)
            if (p[cls0 + c] > best) { best = p[cls0 + c]; bi = c; }
        float sc = p[4] * best;
        if (!(sc >= conf_thr)) continue;
        float *d = det + n * W;
        d[0] = b[0]; d[1] = b[1]; d[2] = b[2]; d[3] = b[3]; d[4] = p[4]; d[5] = best;
        for (int64_t e = 0; e < n_extra; ++e) d[6 + e] = p[5 + e];
        d[W - 1] = (float)bi;
        float off = d[W - 1] * 4096.0f;          /* class offset, fp32 */
        for (int q = 0; q < 4; ++q) obox[4 * n + q] = b[q] + off;
        score[n] = sc;
        src[n] = a;
        ++n;
    }
    int64_t k = yo_nms(obox, score, n, nms_thr, keep);
    for (int64_t i = 0; i < k; ++i) {
        memcpy(rows + i * W, det + keep[i] * W, (size_t)W * sizeof(float));
        if (keep_src) keep_src[i] = src[keep[i]];
    }
    free(det); free(obox); free(score); free(src); free(keep);
    return k;
}
