/*
 * yolo_amd.h -- C ABI of libyolo_amd.so (gfx950 / MI355X).
 *
 * Drop-in boundary for the inference hot path of the luoyang YOLO reference:
 * conv backbone/neck -> detection head -> decode -> confidence filter + per-class NMS.
 * The reference is pure Python over ATen/torchvision; each entry point below names the
 * reference code it replaces (paths relative to the reference repository root).
 *
 * Conventions
 *   - every function returns 0 on success, a negative YAMD_E* code on failure; the message
 *     is available from yamd_last_error() (thread local).  Nothing throws across the ABI.
 *   - all data pointers are DEVICE pointers owned by the caller unless a parameter is
 *     explicitly called *_host.  The library allocates no device memory.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Calls only
 *     enqueue work; they never synchronise.
 *   - activations are NHWC bf16 with an explicit pixel stride in elements ("cstride"), so a
 *     tensor may be a channel slice of a wider concat buffer: element (b,y,x,c) lives at
 *     base[((b*H + y)*W + x)*cstride + c].
 */
#ifndef YOLO_AMD_H
#define YOLO_AMD_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define YAMD_OK 0
#define YAMD_EINVAL (-1)   /* bad argument / unsupported shape */
#define YAMD_EHIP (-2)     /* a HIP runtime call failed */
#define YAMD_ENOSPC (-3)   /* workspace too small */

/* activation fused into the conv epilogue */
#define YAMD_ACT_NONE 0
#define YAMD_ACT_SILU 1    /* x*sigmoid(x): yolov5/YoloV5Body.py:16-23, v7/v8 Conv */
#define YAMD_ACT_LEAKY 2   /* LeakyReLU(0.1): yolov3/DarkNet53.py, v4 neck */
#define YAMD_ACT_MISH 3    /* x*tanh(softplus(x)): yolov4/YoloV4BackBone.py:10-15 */
#define YAMD_ACT_RELU 4    /* v6 RepVGG blocks: yolov6_face/YoloV6FaceBody.py:160 */

const char *yamd_last_error(void);
int yamd_version(void);

/* Optional per-device set-up, for the device that is current on the calling thread: raises the dynamic-LDS limit of
 * every kernel of the library that needs it.  Every entry point does this lazily for the kernels it launches (once per
 * kernel and device, safe from several host threads), so the call is only needed where that lazy step is unwanted:
 * before a region under HIP stream capture, or to take the one-off cost out of the first request.  Call it once per
 * device, with that device current.  No counterpart in the reference (PyTorch initialises its kernels the same lazy
 * way).
 * Stream capture: after this call (and one eager call that sizes the caller's buffers) the entry points record as kernel
 * nodes only and allocate nothing.  ONE launch of the instantiated graph has been checked against the eager calls bit for
 * bit; launching the same executable graph a second time ended in a GPU memory fault on ROCm 7.2 whose cause is not
 * known, so repeated launches of a recorded step are NOT a supported use: issue the calls eagerly (they are
 * asynchronous). */
int yamd_init(void);

/* ------------------------------------------------------------------------------------------
 * Conv + folded BatchNorm + activation (+ residual add) as one NHWC implicit-GEMM MFMA kernel.
 * Replaces nn.Conv2d -> nn.BatchNorm2d -> act of e.g. yolov5/YoloV5Body.py:26-51 (Conv.forward
 * / forward_fuse), the shortcut add of Bottleneck (:80-93) and the bias-only 1x1 head convs
 * (:219-221).  groups=1, dilation=1, zero padding k/2; (k,s) in {(1,1),(3,1),(3,2)}.
 * ---------------------------------------------------------------------------------------- */
typedef struct yamd_conv_desc {
    int32_t ksize, stride;
    int32_t cin;       /* channels of the stored input (multiple of 16; zero-padded weights) */
    int32_t cout;      /* real output channels */
    int32_t act;       /* YAMD_ACT_* */
    int32_t out_f32;   /* 0: y is bf16; 1: y is float (raw head logits) */
    /* filled by yamd_conv_desc_init (kernel tiling; fixes the packed-weight layout) */
    int32_t nrep, ck, mrep, cout_pad;
    int64_t packed_w_elems;   /* bf16 elements of the packed weight blob */
    int32_t nwaves;           /* 4 or 8 wavefronts per workgroup (they share one staged weight tile) */
    float residual_scale;     /* y = act(..) + residual_scale * residual; yamd_conv_desc_init sets 1 (the learned
                               * shortcut weight ``alpha`` of BottleRep, yolov6/YoloV6Body.py:451-472) */
    int32_t stages;           /* 1: one LDS image of the staged input chunk; 2: two, filled alternately (one barrier per
                               * chunk instead of two), a workgroup walks every cout tile of its pixel tile; 3: as 2, one
                               * cout tile per workgroup (the workgroups of a pixel tile share its input through L2) */
} yamd_conv_desc;

int yamd_conv_desc_init(yamd_conv_desc *d, int ksize, int stride, int cin, int cout, int act, int out_f32);

/* Optional: override the tiling chosen by yamd_conv_desc_init (per-layer tuning) before the weights are packed.
 * mrep in {1,2}: 128 or 256 output pixels per workgroup; nrep in {1,2,4}: 32*nrep output channels per workgroup;
 * ck in {16,32,64}: input channels staged per step; nwaves in {4,8}: wavefronts per workgroup, i.e. the pixel tile
 * is nwaves*32*mrep pixels; stages in {1,2,3}: see yamd_conv_desc.  Fails if that kernel is not built.  No
 * counterpart in the reference (MIOpen picks its own algorithm there). */
int yamd_conv_desc_tune(yamd_conv_desc *d, int mrep, int nrep, int ck, int nwaves, int stages);

/* Host-side: fold nothing, just reorder.  w_oihw_host is float [cout][cin_real][k][k] (BN already
 * folded by the caller, cf. utils/torch_utils.py:344-370), bias_host float[cout] or NULL.
 * Writes packed bf16 weights (d->packed_w_elems uint16) and float bias (d->cout_pad). */
int yamd_conv_pack_weights(const yamd_conv_desc *d, const float *w_oihw_host, int cin_real,
                           const float *bias_host, uint16_t *w_packed_host, float *bias_packed_host);

/* y = act(conv(x, w) + bias) (+ residual).  x: (batch,h,w,cin) bf16 NHWC; y: (batch,hout,wout,cout)
 * with hout = (h + 2*(k/2) - k)/s + 1.  residual may be NULL. */
int yamd_conv2d_bias_act(const yamd_conv_desc *d, const void *x, int x_cstride, const void *w_packed,
                         const float *bias_packed, const void *residual, int res_cstride, void *y,
                         int y_cstride, int batch, int h, int w, void *stream);
/* The same conv with its output channels going to two places: couts [0, split_cout) to y, couts [split_cout, cout) to
 * y2 (at channel cout - split_cout).  Used where one conv computes two branches that read the same input (C3's cv1 and
 * cv2, yolov5/YoloV5Body.py:96-111) and their consumers should find each branch densely packed.  bf16 output, no
 * residual; split_cout % 8 == 0. */
int yamd_conv2d_bias_act_split(const yamd_conv_desc *d, const void *x, int x_cstride, const void *w_packed,
                               const float *bias_packed, void *y, int y_cstride, void *y2, int y2_cstride,
                               int split_cout, int batch, int h, int w, void *stream);

/* A 1x1 conv whose input is torch.cat((a, b), 1) with either part optionally behind nn.Upsample(scale_factor=2,
 * mode="nearest"), without building the concatenation or the upsampled tensor:
 *     p5_up_sample = self.up_sample(p5); p4 = torch.cat([p5_up_sample, feat2], 1); p4 = self.conv3_for_up_sample1(p4)
 * (yolov5/YoloV5Body.py:232-235 and :237-240, yolov8/YoloV8Body.py:292-297; the first conv(s) of the C3 / C2f block read
 * the cat).  xa holds the first ca input channels, xb the remaining d->cin - ca; xa_half / xb_half = 1: that tensor is
 * (batch, h/2, w/2, *) and pixel (y, x) reads its pixel (y/2, x/2).  ca and d->cin - ca must be multiples of d->ck.
 * y2 = NULL: all output channels to y; otherwise split as in yamd_conv2d_bias_act_split.  No residual. */
int yamd_conv2d_bias_act_cat2(const yamd_conv_desc *d, const void *xa, int xa_cstride, int ca, int xa_half,
                              const void *xb, int xb_cstride, int xb_half, const void *w_packed,
                              const float *bias_packed, void *y, int y_cstride, void *y2, int y2_cstride,
                              int split_cout, int batch, int h, int w, void *stream);

/* ------------------------------------------------------------------------------------------
 * Layout / resampling helpers of the conv graph.
 * ---------------------------------------------------------------------------------------- */
/* Focus space-to-depth + NCHW fp32 -> NHWC bf16 (yolov5/YoloV5Body.py:71-77).  images: (batch,3,h,w)
 * float NCHW; y: (batch,h/2,w/2,16) bf16, channels [ee RGB, oe RGB, eo RGB, oo RGB, 0,0,0,0] where
 * the first letter is the row parity (e=even, o=odd) and the second the column parity. */
int yamd_focus_pack(const float *images, void *y, int y_cstride, int batch, int h, int w, void *stream);
/* Plain NCHW fp32 -> NHWC bf16 with channels zero-padded to y's 16 (stems without Focus). */
int yamd_nchw_to_nhwc16(const float *images, void *y, int y_cstride, int batch, int cin, int h, int w,
                        void *stream);
/* nn.Upsample(scale_factor=2, mode="nearest") (yolov5/YoloV5Body.py:205) writing into a channel
 * slice: x (batch,h,w,c) -> y (batch,2h,2w,c). c % 8 == 0. */
int yamd_upsample2x(const void *x, int x_cstride, void *y, int y_cstride, int batch, int h, int w, int c,
                    void *stream);
/* Three cascaded 5x5 stride-1 max-pools with -inf padding: y1 = mp5(x), y2 = mp5(y1), y3 = mp5(y2),
 * i.e. the k=5,9,13 pools of SPP (yolov5/YoloV5Body.py:114-132, yolov4/YoloV4BackBone.py:151-162)
 * and the serial pools of SPPF (yolov8/YoloV8Body.py:102-114).  All four tensors are channel slices
 * (c % 8 == 0) and may live in the same concat buffer. h*w <= 4096. */
int yamd_maxpool5_cascade(const void *x, int x_cstride, void *y1, void *y2, void *y3, int y_cstride,
                          int batch, int h, int w, int c, void *stream);
/* The same with 3x3 pools: y1 = mp3(x), y2 = mp3(y1), y3 = mp3(y2), i.e. the k=3,5,7 pools of the face / pose
 * SPPF (yolov5_face/Yolov5FaceBody.py:106-125). */
int yamd_maxpool3_cascade(const void *x, int x_cstride, void *y1, void *y2, void *y3, int y_cstride,
                          int batch, int h, int w, int c, void *stream);
/* nn.MaxPool2d(kernel_size=2, stride=2) of the v7 transition blocks (yolov7_pose/YoloV7PoseBody.py:76-82):
 * x (batch,h,w,c) -> y (batch,h/2,w/2,c); h, w even, c % 8 == 0. */
int yamd_maxpool2(const void *x, int x_cstride, void *y, int y_cstride, int batch, int h, int w, int c,
                  void *stream);
/* Second half of nn.ConvTranspose2d(kernel 2, stride 2) (yolov6_face/YoloV6FaceBody.py:475-492): the transposed
 * conv is run as a 1x1 conv to 4*c channels ordered (dy, dx, c); this scatters x (batch,h,w,4c) to
 * y (batch,2h,2w,c) with y[b,2i+dy,2j+dx,:] = x[b,i,j,(2dy+dx)*c : (2dy+dx+1)*c].  c % 8 == 0. */
int yamd_depth_to_space2(const void *x, int x_cstride, void *y, int y_cstride, int batch, int h, int w, int c,
                         void *stream);
/* Channel-slice copy (batch*h*w pixels, c channels, c % 8 == 0): torch.cat when a producer could
 * not write its slice in place. */
int yamd_copy_channels(const void *x, int x_cstride, void *y, int y_cstride, long long pixels, int c,
                       void *stream);

/* ------------------------------------------------------------------------------------------
 * Head decode.
 * ---------------------------------------------------------------------------------------- */
#define YAMD_DECODE_V5 0    /* yolov5/YoloV5Head.py:23-60 (== yolov7/YoloV7Head.py) */
#define YAMD_DECODE_V3 1    /* yolov3/YoloHead.py:23-65 (v3, v4) */
#define YAMD_DECODE_POSE 2  /* yolov7_pose/YoloV7PoseHead.py:25-81 (== yolov5_pose) */
#define YAMD_DECODE_FACE 3  /* yolov7_face/YoloV7FaceHead.py:25-77 (== yolov5_face) */

/* One pyramid level.  raw: (batch,h,w,na*c) float NHWC head logits.  anchors_wh_host: na*2 floats,
 * anchor sizes in input pixels (host memory, copied into the launch).
 * merge == 0: writes the head module's output, (batch,na,h,w,c) in grid units, to out.
 * merge == 1: additionally applies the level merge of yolov5/YoloV5.py:60-69: box (and point) columns
 *   x stride, rows land at out[b][row_offset + a*h*w + y*w + x][:] of a (batch, rows_per_image, c)
 *   tensor, cx,cy,w,h clamped to [0, img_size-1]; clamp_points additionally clamps the point
 *   columns (yolov7_pose/YoloV7Pose.py:156-159, predict only). */
int yamd_decode_anchor(int kind, const float *raw, int batch, int h, int w, int na, int c,
                       const float *anchors_wh_host, int stride, int point_num, float img_size, int merge,
                       int clamp_points, float *out, long long rows_per_image, long long row_offset,
                       void *stream);

/* All pyramid levels of an anchor head in one launch, merged form only (merge == 1 above): raw[i] is
 * (batch,h_i,w_i,na*c) float NHWC, anchors_wh_host holds n_levels*na*2 floats (level-major), out is
 * (batch, na * sum h_i*w_i, c) with the levels in the given order (yolov5/YoloV5.py:60-69, yolov3/YoloV3.py:129-139,
 * yolov7_pose/YoloV7Pose.py:143-159).  n_levels <= 8, na <= 4. */
int yamd_decode_anchor_levels(int kind, const float *const *raw_host_ptrs, const int *hs_host, const int *ws_host,
                              const int *strides_host, const float *anchors_wh_host, int n_levels, int batch, int na,
                              int c, int point_num, float img_size, int clamp_points, float *out, void *stream);

/* Anchor-free DFL decode, one call for all levels (yolov8/YoloV8Head.py:28-60, YoloV8Body.py:204-223,
 * utils/torch_utils.py:373-394, yolov8/YoloV8.py:48-57 and :118-124).  raw[i]: (batch,h_i,w_i,4*reg_max+nc)
 * float NHWC, channel order [reg 4*reg_max | cls nc].  out: (batch, sum h_i*w_i, 5+nc) =
 * [box, max class score, class scores]; box xyxy (out_cxcywh=0) or cxcywh (1), input pixels. */
int yamd_decode_dfl(const float *const *raw_host_ptrs, const int *hs_host, const int *ws_host,
                    const int *strides_host, int n_levels, int batch, int reg_max, int nc, int out_cxcywh,
                    float *out, void *stream);
/* The DFL decode of YOLOv6 m/l (yolov6/YoloV6Body.py:1322-1343 bbox_decode with use_dfl, :1345-1380; assembled by
 * yolov6/YoloV6.py:41-55 and :62-74): ``bins`` = reg_max + 1 logits per box side whose softmax is projected on
 * 0..reg_max; column 4 of ``out`` is the constant 1.0 the reference puts in place of an objectness.  Otherwise as
 * yamd_decode_dfl. */
int yamd_decode_dfl_unit_conf(const float *const *raw_host_ptrs, const int *hs_host, const int *ws_host,
                              const int *strides_host, int n_levels, int batch, int bins, int nc, int out_cxcywh,
                              float *out, void *stream);

/* Anchor-free decode without DFL plus landmarks, all levels in one call (yolov6_face/YoloV6FaceBody.py:1354-1397
 * bbox_decode/point_decode and :1439-1455, assembled by yolov6_face/YoloV6Face.py:42-58 and :60-74).
 * raw[i]: (batch,h_i,w_i,4+2*point_num+nc) float NHWC, channel order [l,t,r,b | x0,y0,x1,y1.. | cls nc].
 * out: (batch, sum h_i*w_i, 5+2*point_num+nc) = [box, 1.0, landmarks (pixels), sigmoid(cls)]; box xyxy
 * (out_cxcywh=0, predict form) or cxcywh (1, _forward form), input pixels. */
int yamd_decode_ltrb_points(const float *const *raw_host_ptrs, const int *hs_host, const int *ws_host,
                            const int *strides_host, int n_levels, int batch, int point_num, int nc, int out_cxcywh,
                            float *out, void *stream);
/* The same rows for the m/l widths of the YOLOv6 face and pose heads, whose box branch is a DFL distribution
 * (yolov6_face/YoloV6FaceBody.py:1354-1376 bbox_decode with use_dfl, :1426-1429): raw[i] is
 * (batch,h_i,w_i,4*bins+2*point_num+nc), channel order [4 x bins box logits | landmarks | cls]; each box distance is the
 * expectation of softmax(bins logits) over 0..bins-1 (bins = reg_max + 1 = 17). */
int yamd_decode_dfl_points(const float *const *raw_host_ptrs, const int *hs_host, const int *ws_host,
                           const int *strides_host, int n_levels, int batch, int bins, int point_num, int nc,
                           int out_cxcywh, float *out, void *stream);

/* ------------------------------------------------------------------------------------------
 * Confidence filter + per-class greedy NMS for a whole batch; replaces the Python per-image loop
 * of utils/torch_utils.py:101-141 (n_extra = 0), :144-191 (n_extra = 2P) and :194-241 (n_extra = 3P)
 * including its torchvision.ops.box_convert / torchvision.ops.nms calls.
 *   pred      (batch, a, c) float: [b0,b1,b2,b3,obj, n_extra columns, class scores...]
 *   out_rows  (batch, max_out, 7+n_extra) float: [x1,y1,x2,y2,obj,cls_conf, extra..., cls_id], rows
 *             in descending score order (ties: ascending anchor index) exactly as the reference
 *             returns them
 *   out_index (batch, max_out) int32 source anchor index of every kept row, or NULL
 *   out_counts (batch) int32 kept rows per image (0 <=> the reference returns None); rows beyond
 *             max_out are dropped but still counted
 * The IoU test is bit-exact with the torchvision CPU kernel: fp32 arithmetic on the class-offset
 * boxes (offset = cls*4096 added in fp32), strict '>', compared against the double threshold.
 * ---------------------------------------------------------------------------------------- */
size_t yamd_nms_workspace_bytes(int batch, int a, int c, int n_extra);
int yamd_nms_batched(const float *pred, int batch, int a, int c, int in_cxcywh, int n_extra, float img_size,
                     float conf_thr, double iou_thr, float *out_rows, int *out_index, int max_out,
                     int *out_counts, void *workspace, size_t workspace_bytes, void *stream);

/* The predict path's post-processing as one call: yamd_decode_anchor_levels followed by yamd_nms_batched (in_cxcywh = 1),
 * i.e. everything `predict` does after `self.body_net(images)` (yolov5/YoloV5.py:108-127 over yolov5/YoloV5Head.py:23-60 and
 * utils/torch_utils.py:101-141; the pose / face forms :144-241), with the confidence filter run inside the decode kernel
 * on the rows it has just decoded.  Only rows that pass are written to `pred` (the NMS kernels read them there); the others
 * are left as they were, so `pred` is scratch of this call, not a result: callers that want the decoded tensor
 * (`_forward`) use yamd_decode_anchor_levels.  out_rows / out_index / out_counts are those of yamd_nms_batched on the
 * fully decoded tensor, bit for bit.  n_extra follows from the kind: 3 * point_num (pose), 2 * point_num (face), else 0.
 * workspace as for yamd_nms_batched with a = na * sum(h * w). */
int yamd_decode_nms_anchor(int kind, const float *const *raw_host_ptrs, const int *hs_host, const int *ws_host,
                           const int *strides_host, const float *anchors_wh_host, int n_levels, int batch, int na, int c,
                           int point_num, float img_size, int clamp_points, float *pred, float conf_thr, double iou_thr,
                           float *out_rows, int *out_index, int max_out, int *out_counts, void *workspace,
                           size_t workspace_bytes, void *stream);

/* yamd_nms_batched for callers that cannot allocate device memory (code under HIP graph capture, tight serving loops,
 * harnesses that forbid hipMalloc): the same kernels, with the workspace taken from a static device array of the
 * library instead of a caller buffer, so the call allocates and frees nothing.  No counterpart in the reference.
 *   - the array holds yamd_nms_static_capacity_bytes() = 16 MiB on every GPU the library runs on, for the life of the
 *     process; a shape whose yamd_nms_workspace_bytes() exceeds that is refused with YAMD_ENOSPC (16 MiB covers e.g.
 *     8 images x 25 200 anchors; larger batches use yamd_nms_batched with their own workspace);
 *   - all calls on one GPU share that array: calls that overlap in time on one GPU (different streams or host
 *     threads) race; calls ordered on one stream are safe, each clears the counters it needs itself.
 * Inputs, outputs and semantics are those of yamd_nms_batched. */
size_t yamd_nms_static_capacity_bytes(void);
int yamd_nms_batched_static(const float *pred, int batch, int a, int c, int in_cxcywh, int n_extra, float img_size,
                            float conf_thr, double iou_thr, float *out_rows, int *out_index, int max_out,
                            int *out_counts, void *stream);

/* Packs the kept rows of all images back to back for a single device->host copy; the counterpart of
 * the per-image `.detach().cpu().numpy()` of yolov5/YoloV5.py:125-126.  compact: (sum_b min(count_b,
 * max_out), width) float; offsets: (batch+1) int64, offsets[b] = first row of image b, offsets[batch] =
 * total rows. */
int yamd_compact_rows(const float *rows, const int *counts, int batch, int max_out, int width, float *compact,
                      long long *offsets, void *stream);

/* ------------------------------------------------------------------------------------------
 * Inference pre-processing (the step in front of the path): transformer/transformer.py:134-178 ResizeImage
 * (aspect-preserving PIL.Image.resize with BICUBIC, pasted centred on a (128,128,128) canvas) + :204-232 ImageNorm
 * (/255, HWC -> CHW, float32).  img: (h, w, 3) uint8 RGB on the device, rows row_stride bytes apart;
 * out: (3, target_h, target_w) float -- one image of the (B,3,S,S) batch the models take.  The resize follows
 * Pillow's resampling bit for bit (separable, horizontal first, 22-bit fixed-point taps, uint8 between the passes).
 * geom_host (4 ints: resize_w, resize_h, dw, dh) and ratio_host (resize_ratio) receive the bookkeeping RestoreSize
 * (:260-295) needs; either may be NULL.
 * ---------------------------------------------------------------------------------------- */
size_t yamd_letterbox_workspace_bytes(int h, int w, int target_h, int target_w);
int yamd_letterbox_u8(const uint8_t *img, int h, int w, int row_stride, int target_h, int target_w, float *out,
                      int *geom_host, double *ratio_host, void *workspace, size_t workspace_bytes, void *stream);

/* ----------------------------------------------------------------------------------------
 * Detection metric: matching of ground truths to predictions.
 * Replaces DetectMetric._get_batch_statistics (luoyang/model/metric/DetectMetric.py:35-105) with
 * anchor_utils.bbox_iou (luoyang/utils/anchor_utils.py:25-44, the "+1" IoU) and x_y_w_h_to_x_y_x_y (:78-85).
 * rows (batch, max_out, width) / counts (batch): the kept rows yamd_nms_batched wrote (class id in the last column).
 * targets (batch, max_targets, 5) fp32 = normalised cx, cy, w, h, class id; all-zero rows are padding
 * (DetectMetric.py:64); target_counts (batch) may be NULL (= max_targets rows each).
 * true_positive (batch, max_out) uint8: 1 where a ground truth claimed the prediction with IoU >= iou_thr.
 * Ground truths claim in stored order; ties in IoU go to the lower row; float32 arithmetic as numpy does it.
 * ---------------------------------------------------------------------------------------- */
int yamd_detect_match(const float *rows, const int *counts, int batch, int max_out, int width, const float *targets,
                      const int *target_counts, int max_targets, float img_w, float img_h, float iou_thr,
                      uint8_t *true_positive, void *stream);

/* ----------------------------------------------------------------------------------------
 * First conv of a network, straight from the fp32 NCHW images (batch,3,h,w) to NHWC bf16:
 *   YAMD_STEM_K3S1 / YAMD_STEM_K3S2  Conv2d(3, cout, 3, stride 1 / 2, pad 1) + bias + act
 *       (yolov8/YoloV8Body.py stem, yolov5_face/Yolov5FaceBody.py:56, yolov7_pose/YoloV7PoseBody.py stem,
 *        yolov6_face RepVGG stem after re-parameterisation, yolov3/DarkNet53.py conv1, yolov4 first conv)
 *   YAMD_STEM_FOCUS                  Focus (yolov5/YoloV5Body.py:54-77): space-to-depth to 12 channels, then
 *       Conv2d(12, cout, 3, 1, 1) + bias + act, computed as one 6x6 stride-2 conv on the image
 * Replaces yamd_focus_pack / yamd_nchw_to_nhwc16 followed by yamd_conv2d_bias_act for those layers (no padded
 * 16-channel copy of the image is written).  Weights: (cout, 3 or 12, 3, 3) fp32 with BN folded, packed on the host.
 * cout <= 64 and a multiple of 8; images are rounded to bf16 as in the unfused path.
 * ---------------------------------------------------------------------------------------- */
enum { YAMD_STEM_K3S1 = 0, YAMD_STEM_K3S2 = 1, YAMD_STEM_FOCUS = 2 };
int64_t yamd_stem_packed_elems(int mode, int cout);   /* bf16 elements of the packed weights (bias: 64 floats) */
int yamd_stem_pack_weights(int mode, const float *w_host, const float *bias_host, int cout, void *w_packed_host,
                           float *bias_packed_host);
int yamd_stem_conv(int mode, const float *images, int batch, int h, int w, const void *w_packed,
                   const float *bias_packed, int cout, int act, void *y, int y_cstride, void *stream);

/* ----------------------------------------------------------------------------------------
 * The first two layers of a backbone in one launch: a stem (YAMD_STEM_FOCUS or YAMD_STEM_K3S2 as above, or
 * YAMD_STEM_K3S1 with c1 = 32) followed by Conv2d(c1, c2, 3, 2, 1) + bias + act, i.e. `x = self.stem(x)` and the first
 * module of `self.dark2` (yolov5/YoloV5Body.py:139-141,160-161, yolov8/YoloV8Body.py:124-126,143-144), `conv1` and the
 * down-sampling conv of `layer_1` (yolov3/DarkNet53.py:58-62,73-76), the first two convs of CSPDarkNet and of the v7
 * backbone (yolov4/YoloV4BackBone.py, yolov7_pose/YoloV7PoseBody.py).  Same result as yamd_stem_conv into a
 * (batch, h/2, w/2, c1) bf16 tensor followed by yamd_conv2d_bias_act on it (the intermediate is rounded to bf16 in
 * both), but that tensor, the largest activation of the network, is never written to memory.
 * c1 = 32 or 64 (the stem's cout); c2 a multiple of 8, at most 128 for c1 = 32 and 64 for c1 = 64; h and w even.
 * w2: (c2, c1, 3, 3) fp32 with BN folded, packed on the host; bias_packed: 128 floats.  y: (batch, h/4, w/4, *) bf16 after
 * a stride-2 stem, (batch, h/2, w/2, *) after a stride-1 stem.
 * ---------------------------------------------------------------------------------------- */
int64_t yamd_stem_conv2_packed_elems(int c1, int c2);   /* bf16 elements of the second conv's packed weights; 0 = unsupported */
int yamd_stem_conv2_pack_weights(const float *w2_host, const float *bias2_host, int c1, int c2, void *w2_packed_host,
                                 float *bias2_packed_host);
int yamd_stem_conv2(int mode, const float *images, int batch, int h, int w, const void *w1_packed,
                    const float *b1_packed, int c1, int act1, const void *w2_packed, const float *b2_packed, int c2,
                    int act2, void *y, int y_cstride, void *stream);

/* ----------------------------------------------------------------------------------------
 * A residual bottleneck in one launch: y = [x +] act(conv3x3(act(conv1x1(x)))), i.e. `x + self.cv2(self.cv1(x))` of
 * Bottleneck (yolov5/YoloV5Body.py Bottleneck.forward), the residual unit of DarkNet53 (yolov3/DarkNet53.py:33-47:
 * conv 1x1 to half the channels, conv 3x3 back, `x += residual`) and CSPDarkNet's Resblock (yolov4/YoloV4BackBone.py).
 * Same result as yamd_conv2d_bias_act twice (the 1x1 conv's output rounded to bf16 in both, the second call with x as
 * its residual), but that intermediate tensor is neither written nor read back and x is read once.
 * Built for cin = cout with (cin, cmid) in {(32,32), (64,64), (64,32), (128,128), (128,64)}; both convs stride 1 with
 * the same activation.  w1: (cmid, cin, 1, 1), w2: (cout, cmid, 3, 3) fp32 with BN folded; the pack call writes
 * yamd_bottleneck_packed_elems() bf16 values and cmid + cout floats.  shortcut != 0 adds x.  x and y are NHWC bf16,
 * disjoint tensors or different channel slices of one buffer.
 * ---------------------------------------------------------------------------------------- */
int64_t yamd_bottleneck_packed_elems(int cin, int cmid, int cout);   /* 0 = this shape is not built */
int yamd_bottleneck_pack_weights(int cin, int cmid, int cout, const float *w1_host, const float *b1_host,
                                 const float *w2_host, const float *b2_host, void *w_packed_host, float *bias_packed_host);
int yamd_bottleneck(int cin, int cmid, int cout, int act, const void *x, int x_cstride, const void *w_packed,
                    const float *bias_packed, int shortcut, void *y, int y_cstride, int batch, int h, int w, void *stream);

/* ----------------------------------------------------------------------------------------
 * A whole C3 block with one bottleneck in one launch (yolov5/YoloV5Body.py:96-111, C3.forward with n = 1, over
 * Bottleneck :80-93):
 *     y = cv3(cat((m(cv1(x)), cv2(x)), 1)),   m(a) = a + m.cv2(m.cv1(a)) (shortcut) or m.cv2(m.cv1(a))
 * every conv followed by its folded BatchNorm bias and SiLU.  Same result as the layer-by-layer path (1x1 conv to the
 * two branches, yamd_bottleneck, 1x1 conv over the concatenation; every intermediate tensor rounded to bf16 in both,
 * fp32 accumulation in a different order), but x is read once and y written once: the four tensors in between never
 * reach memory.  Built for (cin, ch, cout) = (64, 32, 64), the 160 x 160 stage of YOLOv5-s (dark2), SiLU only.
 * Weights fp32 with BN folded: w_cv1, w_cv2 (ch, cin, 1, 1); w_m1 (ch, ch, 1, 1); w_m2 (ch, ch, 3, 3); w_cv3
 * (cout, 2*ch, 1, 1) with input channels ordered [m | cv2 branch] as the reference's cat.  The pack call writes
 * yamd_c3_packed_elems() bf16 values and 4*ch + cout floats.  x, y: NHWC bf16, disjoint tensors or different channel
 * slices of one buffer.
 * ---------------------------------------------------------------------------------------- */
int64_t yamd_c3_packed_elems(int cin, int ch, int cout);   /* 0 = this shape is not built */
int yamd_c3_pack_weights(int cin, int ch, int cout, const float *w_cv1, const float *b_cv1, const float *w_cv2,
                         const float *b_cv2, const float *w_m1, const float *b_m1, const float *w_m2, const float *b_m2,
                         const float *w_cv3, const float *b_cv3, void *w_packed_host, float *bias_packed_host);
int yamd_c3_fused(int cin, int ch, int cout, int act, const void *x, int x_cstride, const void *w_packed,
                  const float *bias_packed, int shortcut, void *y, int y_cstride, int batch, int h, int w, void *stream);

/* ----------------------------------------------------------------------------------------
 * A model's whole conv graph as one call.  Replaces `out5, out4, out3 = self.body_net(images)`, the module-by-module
 * forward of the backbone + neck + head convs (yolov5/YoloV5.py:54 and :79 over yolov5/YoloV5Body.py:159-187,
 * :223-260; the same statement in every other model class), for callers that do not want one foreign call per layer.
 * A plan is a flat list of the entry points above with every argument resolved: ops[k].kind says which entry point,
 * p[] / i[] hold its pointer / integer arguments in the slots listed beside each kind; batch and image size are the
 * plan's, the image pointer and the stream are given per run.  The plan copies the list (and the conv descriptors in
 * it); it owns no device memory: activation buffers and packed weights stay the caller's and must outlive it.
 * yamd_plan_run only enqueues (one launch per op, in order, on `stream`) and stops at the first op that fails, whose
 * index it stores to *failed_op (may be NULL; -1 on success).  One plan may be run from one host thread at a time.
 * Pointer slots: p[0] input, p[1] packed weights, p[2] packed bias, p[3] residual or NULL, p[4] output (y / y1),
 * p[5] second output (y2), p[6] third output (y3); for YAMD_OP_STEM_CONV2 p[5] / p[6] are the second conv's packed
 * weights / bias.
 * ---------------------------------------------------------------------------------------- */
enum {
    YAMD_OP_CONV = 1,             /* yamd_conv2d_bias_act: desc; i = {x_cstride, res_cstride, y_cstride, h, w} */
    YAMD_OP_CONV_SPLIT = 2,       /* yamd_conv2d_bias_act_split: desc; i = {x_cstride, -, y_cstride, h, w, y2_cstride, split_cout} */
    YAMD_OP_STEM = 3,             /* yamd_stem_conv: i = {mode, cout, act, y_cstride} */
    YAMD_OP_STEM_CONV2 = 4,       /* yamd_stem_conv2: i = {mode, c1, act1, y_cstride, c2, act2} */
    YAMD_OP_FOCUS = 5,            /* yamd_focus_pack: i = {y_cstride} */
    YAMD_OP_NHWC16 = 6,           /* yamd_nchw_to_nhwc16: i = {y_cstride, cin} */
    YAMD_OP_UPSAMPLE2X = 7,       /* yamd_upsample2x: i = {x_cstride, y_cstride, h, w, c} */
    YAMD_OP_MAXPOOL5_CASCADE = 8, /* yamd_maxpool5_cascade: i = {x_cstride, y_cstride, h, w, c} */
    YAMD_OP_MAXPOOL3_CASCADE = 9, /* yamd_maxpool3_cascade: i = {x_cstride, y_cstride, h, w, c} */
    YAMD_OP_MAXPOOL2 = 10,        /* yamd_maxpool2: i = {x_cstride, y_cstride, h, w, c} */
    YAMD_OP_DEPTH_TO_SPACE2 = 11, /* yamd_depth_to_space2: i = {x_cstride, y_cstride, h, w, c (of the output)} */
    YAMD_OP_COPY_CHANNELS = 12,   /* yamd_copy_channels over batch*h*w pixels: i = {x_cstride, y_cstride, h, w, c} */
    YAMD_OP_CONV_CAT2 = 13,       /* yamd_conv2d_bias_act_cat2: desc; p[0] = xa, p[3] = xb, p[5] = y2 or NULL;
                                   * i = {xa_cstride, xb_cstride, y_cstride, h, w, y2_cstride, split_cout, ca, xa_half | xb_half << 1} */
    YAMD_OP_BOTTLENECK = 14,      /* yamd_bottleneck: i = {x_cstride, cin, y_cstride, h, w, cmid, cout, act, shortcut} */
    YAMD_OP_C3 = 15               /* yamd_c3_fused: i = {x_cstride, cin, y_cstride, h, w, ch, cout, act, shortcut} */
};
typedef struct yamd_plan_op {
    int32_t kind;
    int32_t i[9];
    void *p[7];
    yamd_conv_desc desc;          /* YAMD_OP_CONV / YAMD_OP_CONV_SPLIT only */
} yamd_plan_op;
typedef struct yamd_plan yamd_plan;
int yamd_plan_create(const yamd_plan_op *ops, int n_ops, int batch, int img_h, int img_w, yamd_plan **plan_out);
int yamd_plan_run(const yamd_plan *plan, const float *images, void *stream, int *failed_op);
int yamd_plan_size(const yamd_plan *plan);      /* number of ops (0 for NULL) */
void yamd_plan_destroy(yamd_plan *plan);

/* ----------------------------------------------------------------------------------------
 * Measurement helpers for the benchmark's roofline denominators; no counterpart in the reference and not on the
 * detection path.
 *   yamd_microbench_mfma_bf16: `blocks` workgroups of 4 wavefronts, each wavefront issues 4*iters
 *       v_mfma_f32_32x32x16_bf16 (32768 FLOP each) on register operands taken from `operands` (4 KB of bf16, device);
 *       `sink` receives blocks*256 floats.  FLOP = blocks * 4 * 4 * iters * 32768.
 *   yamd_microbench_copy: dst[i] = src[i] with 16-byte accesses, every workgroup on contiguous 16 KB pieces;
 *       variant 0 / 1: 4 / 8 pieces in flight per lane, 2 / 3: the same with non-temporal accesses; bytes % 16 == 0.
 *       Traffic = 2 * bytes.
 * ---------------------------------------------------------------------------------------- */
int yamd_microbench_mfma_bf16(const void *operands, float *sink, int blocks, int iters, void *stream);
int yamd_microbench_copy(const void *src, void *dst, size_t bytes, int variant, void *stream);
/*   yamd_microbench_clock_probe: one wavefront writes n_samples pairs (shader clock counter of its XCD, constant-rate
 *       counter) to `samples` (2*n_samples uint64, device), sleeping naps * 127 * 64 cycles between pairs; launched on a
 *       second stream it records the shader clock the device holds under the load of the first.  n_samples <= 65536,
 *       naps <= 64: the kernel always ends by itself. */
int yamd_microbench_clock_probe(unsigned long long *samples, int n_samples, int naps, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* YOLO_AMD_H */
