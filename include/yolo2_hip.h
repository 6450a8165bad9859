/* libyolo2_hip.so -- C ABI of the MI355X (gfx950) Darknet-19 / YOLO grid-detector path.
 *
 * The reference (wenxichen/tensorflow_yolo2) has no FFI: its boundary is a set of
 * Python functions that build TF1 graph nodes.  Each entry point below names the
 * reference interface it replaces (paths relative to the reference root).  The
 * Python mirrors of those functions (tensorflow_yolo2_amd/yolo2_nets/) bind
 * these symbols with ctypes; INTEGRATION.md shows the stub.
 *
 * Conventions: plain C types only; every function returns 0 on success or a
 * negative error code (message: y2_last_error()); the CALLER owns every device
 * buffer (parameters, gradients, BN state, workspace, inputs, outputs); no hidden
 * device allocation; all work is enqueued on the hipStream_t passed as `stream`
 * (void*, NULL = default stream) and is asynchronous w.r.t. the host; one context
 * per thread/device (thread-compatible, not thread-safe).
 * Tensors are NHWC fp32 at the boundary (as the reference feeds TF); filters HWIO.
 */
#ifndef YOLO2_HIP_H
#define YOLO2_HIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif
/* the library is built with -fvisibility=hidden: exactly the declarations below are exported */
#pragma GCC visibility push(default)

typedef struct y2_ctx y2_ctx;

/* arithmetic type of the MFMA contractions / stored activations */
#define Y2_F32 0  /* exact-f32 MFMA (v_mfma_f32_32x32x2_f32): parity mode        */
#define Y2_F16 1  /* fp16 operands, fp32 accumulate                              */
#define Y2_BF16 2 /* bf16 operands, fp32 accumulate                              */
/* Round 5: the reference's precision on the FAST matrix pipe.  The reference is fp32 end to end
 * (src/yolo2_nets/darknet.py:10-46, tf.float32 placeholders src/pascal/pascal_train_darknet.py:34-36).  Every MFMA
 * operand (activation, filter, dY) is a pair of halves hi = f16(v), lo = f16(v - hi) in two planes of its channel row
 * (4 bytes per element, as fp32); a product is hi*hi + lo*hi + hi*lo on v_mfma_f32_*_f16 with fp32 accumulation
 * (~22 mantissa bits, a third of the f16 rate instead of the exact-f32 MFMA's sixteenth); conv outputs, gradients
 * with respect to activations, statistics, batch norm, loss and optimizer are fp32 as in Y2_F32; the 3-channel image
 * layer runs in exact fp32.  Gradients ride on the f16 loss scale (y2_set_options) with the f16 mode's overflow guard.
 * Range: the filter planes hold 64 * w in f16 (a power-of-two pre-scale that keeps typical weights' lo plane out of the
 * subnormals), so |w| must stay below 1024; activations and dY have f16's range. */
#define Y2_F16X2 3
/* Round 6: Y2_F16X2 forward, single-product backward.  Tensors, forward pass, batch norm, loss and optimizer are exactly
 * Y2_F16X2's (so every forward DECISION -- leaky branch, pool arg-max, responsible box -- is the reference-precision one);
 * the two backward contractions of tf.gradients (Conv2DBackpropInput / Conv2DBackpropFilter behind
 * src/pascal/pascal_train_darknet.py:49-51) read the hi planes of dY, W and x only: one f16 MFMA per product with fp32
 * accumulation.  With the decisions fixed the backward pass is linear in dY, so the f16 operand rounding (2^-11 relative
 * per element, random) is not amplified: gradients stay within the 1e-3 tolerance of the whole-step gates
 * (tests/test_gpu_f16x2f.py), at roughly a third of the backward cost of Y2_F16X2.  y2_ctx_create and the op-level
 * y2_conv2d / y2_conv2d_backward accept it. */
#define Y2_F16X2F 4

#define Y2_TAIL_NONE 0    /* output = last layer activation [N,Ho,Wo,Cout]       */
#define Y2_TAIL_AVGPOOL 1 /* + average_pooling2d(k,k) + reshape -> [N,Cout]      */

#define Y2_OK 0
#define Y2_ERR_ARG -1
#define Y2_ERR_HIP -2
#define Y2_ERR_STATE -3

const char* y2_last_error(void);
int y2_version(void);

/* Layer list of the reference networks, 4 ints per layer (filter_size, in_chl,
 * out_chl, maxpool_after).  kind 0: darknet19_core (src/yolo2_nets/darknet.py:126-179),
 * kind 1: core + darknet19_detection(output_filter) (darknet.py:182-201),
 * kind 2: darknet19 classifier (darknet.py:61-123).  Returns the layer count. */
int y2_darknet19_spec(int kind, int output_filter, int* spec, int max_layers);

/* ---- network context (replaces the TF graph + variables built by
 *      conv_bn_layer, src/yolo2_nets/darknet.py:32-46) ------------------------
 * spec: 4 ints per layer (filter_size, in_chl, out_chl, after).  after = 0: nothing; 1: tf.nn.max_pool(2, 2, 'SAME')
 * behind the activation (darknet.py:24-25,45); 2 (round 5, inner layers on even maps): SUBSAMPLE -- the layer is slim's
 * conv2d_same(stride=2) / subsample (src/slim_dir/nets/resnet_utils.py:60-122): its stride-1 output at even rows and
 * columns, batch norm over those positions only (the 3x3 convolution of a stride-2 bottleneck unit, resnet_v1.py:99-112) */
int y2_ctx_create(y2_ctx** out, const int* spec, int num_layers, int core_layers, int tail, int tail_k,
                  int batch, int height, int width, int dtype);
void y2_ctx_destroy(y2_ctx* ctx);
/* A GROUP of bound contexts on one flat parameter buffer (round 5: the 20 stacks of the ResNet swap) re-packs its filters
 * in ONE launch instead of one per context.  y2_pack_group_table: once, outside any stream capture (a synchronous copy) --
 * the concatenated pack tables into caller-owned device memory (table_bytes >= 128 * layers is ample); returns the layer
 * and block counts.  y2_pack_group_run: the launch; afterwards the contexts' packed copies are current (the lazy pack of
 * their next y2_forward is skipped).  Contexts with a 3-channel first layer pack on their own. */
int y2_pack_group_table(y2_ctx** ctxs, int n, void* table_dev, size_t table_bytes, int* nlayers, int* blocks);
int y2_pack_group_run(y2_ctx** ctxs, int n, const void* table_dev, int nlayers, int blocks, void* stream);
int y2_num_layers(const y2_ctx* ctx);
/* info: k, cin, cout, pool, H, W (conv input = output spatial size), Ho, Wo */
int y2_layer_info(const y2_ctx* ctx, int layer, int info[8]);
/* trainable floats in reference creation order per layer: W (HWIO), b, gamma, beta */
size_t y2_param_count(const y2_ctx* ctx);
/* BN moving statistics per layer: moving_mean, moving_variance */
size_t y2_state_count(const y2_ctx* ctx);
/* off[0..3]: W, b, gamma, beta offsets (floats) into params/grads; off[4..5]: moving_mean, moving_variance into state */
int y2_param_offsets(const y2_ctx* ctx, int layer, size_t off[6]);
int y2_output_shape(const y2_ctx* ctx, int shape[4]);
size_t y2_workspace_bytes(const y2_ctx* ctx, int training);
int y2_bind(y2_ctx* ctx, float* params, float* grads, float* state, void* workspace, size_t workspace_bytes,
            int training, void* stream);
/* loss-scale applied to half-precision gradients (1 = none); moving-variance Bessel switch */
int y2_set_options(y2_ctx* ctx, float grad_scale, int bessel_moving_var);
/* Round 4.  Per-layer activation slopes (max(slope*z, z): 0.1 = the reference's leaky ReLU, src/yolo2_nets/darknet.py:5,45;
 * 0 = ReLU, 1 = no activation) and the batch-norm constants of the stack (defaults: tf.layers.batch_normalization's
 * eps 1e-3 / momentum 0.99, darknet.py:39-44).  For stacks that restate slim's resnet_v1 bottleneck
 * (src/slim_dir/nets/resnet_v1.py:99-112; arg scope src/slim_dir/nets/resnet_utils.py:230-257: conv2d without bias +
 * batch_norm(decay 0.997, epsilon 1e-5) + ReLU, the unit's last conv without activation): slopes {0, 0, 1}, eps 1e-5,
 * momentum 0.997, zero_bias_grad = 1 (the bias slots stay in the flat parameter layout; the caller keeps them at zero
 * and they receive no gradient).  slopes == NULL keeps the current slopes; the 3-channel image layer only takes 0.1. */
int y2_set_layer_options(y2_ctx* ctx, const float* slopes, int num_layers, float bn_eps, float bn_momentum,
                         int zero_bias_grad);
/* weight_variable / bias_variable / BN initial values (darknet.py:10-17): truncated
 * normal(0.1) re-drawn beyond 2 sigma, 0.1, gamma 1, beta 0, moving 0 / 1 */
int y2_init_params(y2_ctx* ctx, uint64_t seed, void* stream);
/* call after params were changed from outside (load / optimizer on a foreign buffer) */
int y2_params_changed(y2_ctx* ctx);

/* darknet19_core / darknet19_detection / darknet19 forward
 * (darknet.py:61-201): images [N,H,W,3] fp32 -> out (shape: y2_output_shape) */
/* update_moving: apply the momentum-0.99 moving-statistics update of the training-mode layers inside this
 * forward (the train step); 0 = leave them alone, as a TF run that does not fetch train_op / UPDATE_OPS does
 * (pascal_detect_darknet.py:61: the head's default is_training=True normalises with batch statistics at
 * detect time, but nothing updates).  y2_update_moving_stats applies the skipped update afterwards, once. */
int y2_forward(y2_ctx* ctx, const float* images, int is_training_core, int is_training_head, int update_moving,
               float* out, void* stream);
/* Round 4.  The stack as the residual branch of a bottleneck unit (src/slim_dir/nets/resnet_v1.py:99-112: output =
 * tf.nn.relu(shortcut + residual)): out = max(join + stack(images), 0), join [like out] fp32 -- the values of y2_forward
 * followed by y2_add_relu, bit for bit; the branch output is never stored.  y2_backward* then takes
 * y2_add_relu_backward's result (the gradient of the branch output), as after the two-call form. */
int y2_forward_join(y2_ctx* ctx, const float* images, const float* join, int is_training_core, int is_training_head,
                    int update_moving, float* out, void* stream);
/* y2_forward fed with what image_read holds BEFORE its float conversion (src/img_dataset/pascal_voc.py:60-67:
 * cv2.imread + cv2.resize give uint8 BGR): images_u8 [N,H,W,3] uint8; the conversion
 * image.astype(float32) / 255.0 * 2.0 - 1.0 (pascal_voc.py:63-64, same fp32 operation order) runs inside the
 * input pack kernel -- a fed training loop uploads 1 byte per value instead of 4. */
int y2_forward_u8(y2_ctx* ctx, const uint8_t* images_u8, int is_training_core, int is_training_head,
                  int update_moving, float* out, void* stream);
int y2_update_moving_stats(y2_ctx* ctx, void* stream);
/* TF autodiff of the stack (tf.train.*Optimizer().minimize, pascal_train_darknet.py:49-51):
 * dout has the output's shape; gradients are written to the bound `grads` buffer
 * for layers [layer_lo, layer_hi) walking downwards; call with (0, num_layers)
 * for the whole net, or in slices to overlap the all-reduce of finished layers. */
int y2_backward(y2_ctx* ctx, const float* dout, int layer_lo, int layer_hi, void* stream);
/* All layers in one pass, with an event pair recorded when every layer >= mark_layers[k] is complete (data
 * parallelism: the all-reduce of a gradient slice starts behind y2_wait_mark on its own stream while the
 * pass continues; no join on `stream` until the end, unlike one y2_backward call per slice). */
int y2_backward_marks(y2_ctx* ctx, const float* dout, int n_marks, const int* mark_layers, void* stream);
int y2_wait_mark(y2_ctx* ctx, int k, void* stream);
/* y2_backward over all layers that also writes the gradient with respect to the stack's input, fp32 NHWC
 * [N,H,W,in_chl of layer 0]: for stacks composed into larger graphs (the YOLOv2 detector of the north star: 13x13
 * stack and head behind the passthrough concat).  The 3-channel image layer has no input gradient. */
int y2_backward_input(y2_ctx* ctx, const float* dout, float* dinput, void* stream);
/* copy a layer's saved activation for tests.  what: 0 = bordered input of `layer` (post BN+leaky+pool of the layer
 * below) [N,H,W,in_chl]; 1 = conv output [N,H,W,out_chl]; 2 = dy * grad_scale [N,H,W,out_chl]; 3 = the per-channel
 * constants the last forward normalised the layer with, [4][out_chl]: mean, 1/sqrt(var + eps), scale, shift */
int y2_debug_read(y2_ctx* ctx, int layer, int what, float* dst, void* stream);

/* Optional measurement aid: bracket every kernel launch of y2_forward / y2_backward with
 * HIP events on the launch stream (the reference only has utils/timer.py wall clocks).
 * Categories: 0 conv fwd (implicit GEMM), 1 conv1 fwd, 2 dgrad, 3 wgrad, 4 conv1 wgrad,
 * 5 BN fwd passes, 6 BN bwd passes, 7 pack/convert.  on = 1: every launch, serialised
 * (no side stream); on = 2: only the MFMA convolution launches (categories 0, 2, 3), streams as in production;
 * on = 0 stops recording but keeps the records, on = 3 resumes mode 2 without clearing (sampling some steps).  collect() waits for the events. */
int y2_profile_enable(y2_ctx* ctx, int on);
int y2_profile_collect(y2_ctx* ctx, double* ms_by_category, int* launches_by_category, int ncat);
/* length of the union of the [start, end] intervals of the launches whose category bit is set in cat_mask
 * (weight gradients run on a side stream beside the dgrads: summing durations would count shared time twice).
 * Call before y2_profile_collect (which resets). */
int y2_profile_busy(y2_ctx* ctx, int cat_mask, double* busy_ms, int* launches);
/* ms[num_layers][8]: the same records per layer and category (does not reset) */
int y2_profile_layers(y2_ctx* ctx, double* ms);

/* ---- get_loss / get_iou / show_yolo_detection (src/yolo2_nets/net_utils.py:222-439) */
size_t y2_yolo_loss_workspace_bytes(int batch, int S);
/* loss[5] = class, object, noobject, coord, total; dnet may be NULL */
int y2_yolo_loss(const float* net, const float* labels, int num_class, int batch, float image_size, int S, int B,
                 float lambda_coord, float lambda_noobj, float* loss, float* ious, float* object_mask,
                 float* dnet, void* workspace, void* stream);
int y2_get_iou(const float* boxes1, const float* boxes2, float* iou, int n_boxes, void* stream);
/* det[(cell*B+b)*8 + {keep, upper_left_x, upper_left_y, w, h, class, cell_row, cell_col}] */
int y2_decode_detections(const float* predict, int S, int B, int num_class, int im_w, int im_h,
                         float object_thresh, int* det, float* conf, void* stream);
/* sparse_softmax_cross_entropy_with_logits + reduce_mean (imagenet_train_darknet.py:51-53) */
int y2_softmax_cross_entropy(const float* logits, const int* labels, int batch, int classes, float* loss,
                             float* dlogits, void* stream);

/* accuracy = reduce_mean(cast(equal(argmax(logits, 1), labels))) (imagenet_train_darknet.py:60-61; ties: the
 * smallest index, as tf.argmax) */
int y2_accuracy(const float* logits, const int* labels, int batch, int classes, float* accuracy, void* stream);

/* ---- YOLOv2 pieces named by the north star that the reference does NOT contain (SURVEY §8 a-x1, a-x2):
 *      no reference interface to cite; specification = oracle/ext_ref.py of this repo. ------------- */
/* standalone tf.nn.max_pool(2, 2, 'SAME') on fp32 NHWC (reference darknet.py:24-25) and its gradient (first
 * maximum in row-major window order); the executor pools inside its BN pass, composed graphs use this op */
int y2_maxpool2x2(const float* x, float* y, int N, int H, int W, int C, void* stream);
int y2_maxpool2x2_backward(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* reorg / space-to-depth: forward x [N,H,W,C] -> y [N,H/s,W/s,s*s*C], channel ((h%s)*s + w%s)*C + c;
 * forward = 0: the inverse permutation (x coarse -> y [N,H,W,C]), i.e. the gradient.  Bit-exact copies. */
int y2_reorg(const float* x, float* y, int N, int H, int W, int C, int stride, int forward, void* stream);
/* passthrough: out [N,H,W,4*Cf+Cc] = concat(reorg2(fine [N,2H,2W,Cf]), coarse [N,H,W,Cc]) and its gradient */
int y2_passthrough_concat(const float* fine, const float* coarse, float* out, int N, int H, int W, int Cf, int Cc,
                          void* stream);
int y2_passthrough_concat_backward(const float* dout, float* dfine, float* dcoarse, int N, int H, int W, int Cf,
                                   int Cc, void* stream);
/* dst += src on fp32 buffers: the gradient of a tensor with two consumers (the 26x26x512 activation feeds the pool
 * and the passthrough) */
int y2_accumulate(float* dst, const float* src, size_t n, void* stream);
/* Round 4.  The join of a bottleneck unit, output = tf.nn.relu(shortcut + residual) (src/slim_dir/nets/resnet_v1.py:112),
 * and its backward g = (dout + dout2) * [out > 0] (the gradient of both addends; dout2 nullable: the incoming gradient
 * as the two branch gradients of the unit above, never summed in memory); fp32 tensors of n elements, 16-byte aligned. */
int y2_add_relu(const float* a, const float* b, float* out, size_t n, void* stream);
int y2_add_relu_backward(const float* dout, const float* dout2, const float* out, float* g, size_t n, void* stream);
/* Round 5.  LINKED stacks: the stride-1 bottleneck units of the ResNet swap (src/slim_dir/nets/resnet_v1.py:99-112;
 * src/pascal/pascal_train_resnet.py:37-50) hand activations and gradients to each other in the arithmetic type instead of
 * through fp32 NHWC tensors (round 4: a cast, a pack and a convert pass on both sides of every join).
 *   y2_bordered_bytes: size of a zero-bordered tensor [N][H+1][W+1][C] of `dtype` with its guard bands, and the byte
 *     offset of cell 0 inside it; the caller allocates it ZEROED once (borders and guards are never written).
 *   y2_link: tensors of the NEXT y2_forward / y2_backward calls of this context (all nullable; pointers address cell 0):
 *     x_bordered   layer 0's input is this tensor (y2_forward's `images` may then be NULL; no pack pass)
 *     out_bordered the last layer's activation is written here (the consumer's input; y2_forward's `out` may be NULL)
 *     join_bordered / join_self   the stored value is relu(join + stack(x)) with the join read from a bordered tensor of
 *                  the output's geometry, or from the stack's own layer-0 input (identity shortcut); either output form
 *     dout_t       [M][out_chl] of T: d loss / d (pre-join output) * grad_scale (y2_backward's `dout` may be NULL)
 *     dx_t         [M][in_chl] of T: receives d loss / d input * grad_scale (beside, or instead of, y2_backward_input's fp32)
 *   y2_join_backward: g = (d1 + d2) * [out > 0], all of T except d2 when d2_f32 (the g of a run's top unit, fp32);
 *     out bordered (the unit's output as its consumer holds it), d1 / d2 / g [M][C]. */
size_t y2_bordered_bytes(int N, int H, int W, int C, int dtype, size_t* cell0_offset);
int y2_link(y2_ctx* ctx, void* x_bordered, void* out_bordered, const void* join_bordered, int join_self, const void* dout_t,
            void* dx_t);
int y2_join_backward(int dtype, const void* out_bordered, const void* d1, const void* d2, int d2_f32, void* g, int N, int H,
                     int W, int C, void* stream);
/* ... with a STRIDE-2 unit above (round 5: the last unit of blocks 1-3 inside a linked run).  y2_subsample_bordered:
 * resnet_utils.subsample(x, 2) (resnet_utils.py:60-75) from a bordered tensor [N][H+1][W+1][C] of T into one of half the
 * size -- that unit's identity shortcut, joined in its last apply pass (y2_link join_bordered).  y2_join_backward_s2: as
 * y2_join_backward, but d2 -- the gradient of that shortcut -- lives on the upper unit's OUTPUT grid [N*H/2*W/2][C] and
 * reaches the even rows / columns of this [N*H*W] grid only. */
int y2_subsample_bordered(int dtype, const void* src_bordered, void* dst_bordered, int N, int H, int W, int C, void* stream);
int y2_join_backward_s2(int dtype, const void* out_bordered, const void* d1, const void* d2, int d2_f32, void* g, int N, int H,
                        int W, int C, void* stream);
/* scores [rows][classes] -> best score and class index per row (the class choice in front of the NMS of the YOLOv2
 * detector; ties: smallest index, as np.argmax in net_utils.py:418) */
int y2_class_argmax(const float* scores, float* best, int* cls, int rows, int classes, void* stream);
/* x *= s: the loss scale in front of a half-precision backward pass of a composed graph (no reference counterpart:
 * the reference runs fp32) */
int y2_scale(float* x, size_t n, float s, void* stream);
/* anchor decode: net [N,S,S,B,5+C] (tx,ty,tw,th,to,classes), anchors [B][2] in cell units ->
 * boxes [N,S*S*B,4] (cx,cy,w,h relative to the image), scores [N,S*S*B,C] = sigmoid(to)*softmax(classes) */
int y2_decode_anchors(const float* net, const float* anchors, float* boxes, float* scores, int N, int S, int B, int C,
                      void* stream);
/* per-image greedy NMS over K <= 4096 candidates: score descending (ties: lower index first), candidates
 * below score_thresh dropped, a kept box suppresses later boxes with IoU > iou_thresh (same class id only
 * when class_aware).  keep [N][max_out] original indices (-1 padded), count [N].  Bit-exact vs the spec. */
int y2_nms(const float* boxes, const float* scores, const int* classes, int N, int K, float iou_thresh,
           float score_thresh, int max_out, int class_aware, int* keep, int* count, void* stream);

/* YOLOv2 anchor-box loss, forward + gradient (specification: oracle/ext_ref.py yolov2_loss): net [N,S,S,B,5+C]
 * raw outputs, labels = the reference's label grid [N,S,S,5+C] (one box per cell, img_dataset/pascal_voc.py:146-163),
 * anchors [B][2] in cell units, scales = {coord, object, noobject, class, iou_thresh} (NULL: 1, 5, 1, 1, 0.6).
 * loss[5] = coord, object, noobject, class, total (mean over the batch); dnet (nullable) same shape as net. */
size_t y2_yolov2_loss_workspace_bytes(int batch);
int y2_yolov2_loss(const float* net, const float* labels, const float* anchors, int batch, int S, int B, int num_class,
                   float image_size, const float* scales, float* loss, float* dnet, void* workspace, void* stream);

/* ---- ResNet-50 backbone swap (src/yolo2_nets/tf_resnet.py:12-32, src/pascal/pascal_train_resnet.py:37-50):
 *      graph-level operators on fp32 NHWC tensors that the conv-BN-leaky stacks do not have.  The 1x1 / 3x3
 *      convolutions of the bottleneck units (slim_dir/nets/resnet_v1.py:68-112) go through y2_conv2d /
 *      y2_conv2d_backward; a stride-2 unit = the stride-1 convolution + y2_subsample, slim's own definition of
 *      conv2d_same (slim_dir/nets/resnet_utils.py:77-122). --------------------------------------------------- */
/* slim.batch_norm (resnet_arg_scope: decay 0.997, epsilon 1e-5, scale; resnet_utils.py:230-257) on [rows][channels],
 * + optional residual add (before the activation) + optional ReLU: y = act(gamma (x - mean) / sqrt(var + eps) + beta
 * + residual).  is_training: batch statistics (saved in save_mean / save_var for the backward pass), moving
 * statistics updated when update_moving; else the moving statistics normalise. */
int y2_batch_norm_forward(const float* x, const float* residual, float* y, size_t rows, int channels, const float* gamma,
                          const float* beta, float* moving_mean, float* moving_var, float* save_mean, float* save_var,
                          float eps, float decay, int is_training, int update_moving, int relu, void* stream);
/* dx, dgamma, dbeta (and dresidual = the gradient entering before the activation, nullable) */
int y2_batch_norm_backward(const float* dy, const float* y, const float* x, float* dx, float* dresidual, size_t rows,
                           int channels, const float* gamma, const float* save_mean, const float* save_var, float eps,
                           int is_training, int relu, float* dgamma, float* dbeta, void* stream);
/* resnet_utils.subsample = max_pool2d([1,1], stride=factor) (resnet_utils.py:60-75): forward x [N,H,W,C] ->
 * y [N,ceil(H/f),ceil(W/f),C]; forward = 0: x is the gradient at the coarse grid, y the gradient at [N,H,W,C] */
int y2_subsample(const float* x, float* y, int N, int H, int W, int C, int factor, int forward, void* stream);
/* slim.max_pool2d(net, [3,3], stride=2) with padding 'SAME' (resnet_v1.py:198) and its gradient */
int y2_maxpool3x3s2(const float* x, float* y, int N, int H, int W, int C, void* stream);
int y2_maxpool3x3s2_backward(const float* x, const float* dy, float* dx, int N, int H, int W, int C, void* stream);
/* root block: conv2d_same(net, 64, 7, stride=2) on the image [N,H,W,3], filter HWIO [7][7][3][Cout] (resnet_v1.py:197) */
int y2_conv7x7s2(const float* x, const float* w, float* y, int N, int H, int W, int Cout, void* stream);
/* the same with the arithmetic type of the model (round 5): Y2_F16 / Y2_BF16 run it on the matrix pipe (operands rounded to
 * the type, fp32 accumulation and output; Cout <= 64, a multiple of 4), any other dtype is y2_conv7x7s2 */
int y2_conv7x7s2_t(const float* x, const float* w, float* y, int N, int H, int W, int Cout, int dtype, void* stream);
int y2_conv7x7s2_backward_filter_t(const float* x, const float* dy, float* dw, int N, int H, int W, int Cout, int dtype,
                                   void* stream);
int y2_conv7x7s2_backward_filter(const float* x, const float* dy, float* dw, int N, int H, int W, int Cout, void* stream);
/* slim.fully_connected (pascal_train_resnet.py:41-46; the flattened 7x7x2048 features -> 4096 -> S*S*(5B+C)) for a
 * batch of 1..128 rows: y [rows][out] = act(x [rows][in] * w [in][out] + bias), bias may be NULL, in_features a multiple
 * of 16.  One pass over the fp32 weights, converted in registers to `dtype` (0 fp32, 1 f16, 2 bf16) for the matrix
 * cores, fp32 accumulation and fp32 results.  Backward: dx [rows][in] = dy * w^T (NULL: skipped),
 * dw [in][out] = x^T * dy (NULL: skipped); dy is the gradient at the pre-activation (y2_bias_relu_backward). */
int y2_fully_connected(const float* x, const float* w, const float* bias, float* y, int rows, int in_features,
                       int out_features, int relu, int dtype, void* stream);
int y2_fully_connected_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, int rows,
                                int in_features, int out_features, int dtype, void* stream);
/* Round 5: the weight gradient of a fully connected layer fused with the guarded Adam update of that weight -- dW is
 * never stored (tf.train.AdamOptimizer(0.0005).minimize over yolo_fc1, src/pascal/pascal_train_resnet.py:41-50: 1.64 GB of
 * weights; writing, scanning and re-reading its gradient were 3 of the 10 passes over that size per step).  x [rows,in],
 * dz [rows,out] (gradient at the pre-activation, times the loss scale), w / m / v [in,out]; ctrl: the control block of
 * y2_adam_step_guarded AFTER that call advanced it for this step (found_inf: nothing moves; lr_t applies).  The caller's
 * overflow scan covers every OTHER gradient (the bias gradient of the same layer is the column sum of dz: a non-finite
 * dz is seen there; round 6: y2_range_check on x and dz makes the guard of THIS product explicit).  rows <= 256 in the 16-bit
 * types (128 in fp32): the batch -- or, data parallel, the batches of all replicas gathered: every rank then applies the
 * identical update from the identical operands instead of all-reducing a 1.64 GB gradient (yolo2_nets/tf_resnet.py). */
int y2_fc_adam_apply_guarded(const float* x, const float* dz, float* w, float* m, float* v, int rows, int in_features,
                             int out_features, int dtype, const void* ctrl, float beta1, float beta2, float eps,
                             float grad_mult, void* stream);
/* slim.fully_connected tail (pascal_train_resnet.py:41-46): y <- act(y + bias) in place; backward dz = dy [y > 0],
 * dbias = column sums; tf.nn.dropout(x, keep_prob) with a mask that is a function of (seed, index) */
int y2_bias_relu(float* y, const float* bias, size_t rows, int channels, int relu, void* stream);
int y2_bias_relu_backward(const float* dy, const float* y, float* dz, float* dbias, size_t rows, int channels, int relu,
                          void* stream);
int y2_dropout(const float* x, float* y, size_t n, float keep_prob, uint64_t seed, void* stream);
/* the same with the seed read from device memory at run time: a train step captured in a HIP graph replays with a new
 * mask every time (the host increments *seed between replays, or a captured kernel does) */
int y2_dropout_dev(const float* x, float* y, size_t n, float keep_prob, const uint64_t* seed, void* stream);

/* ---- optimizers on flat buffers (pascal_train_darknet.py:51, imagenet_train_darknet.py:58) */
int y2_adam_step(float* params, float* m, float* v, const float* grads, size_t n, int step, float lr,
                 float beta1, float beta2, float eps, float grad_mult, void* stream);
int y2_momentum_step(float* params, float* accum, const float* grads, size_t n, float lr, float momentum,
                     float grad_mult, void* stream);
/* The same updates guarded against half-precision gradient overflow (dynamic loss scaling; the fp32
 * reference cannot overflow): ctrl = 8 zero-initialised 32-bit device words {found_inf, step, skipped, -,
 * lr_t, ...}.  A scan sets ctrl.found_inf: y2_grad_check reads the context's SENTINEL ranges of its bound gradient
 * buffer (b / gamma / beta of every layer and one row of the first filter, ~30 k floats: every non-finite value of the
 * backward pass reaches them -- an inf / NaN in a layer's incoming gradient makes the channel sums dbeta
 * non-finite, a dy that overflows at its own f16 store is an operand of the dgrad below it, and the first layer's
 * dy feeds its filter gradient directly); y2_grad_check_full reads every element of any buffer.
 * The guarded step then skips as a whole when found_inf is set (params, slots and the device-side step
 * counter untouched, ctrl.skipped += 1); otherwise ctrl.step advances and Adam's lr_t is computed on the
 * device for it.  No host synchronisation. */
int y2_grad_check(y2_ctx* ctx, void* ctrl, void* stream);
/* the same scan OR-ed into ctrl.found_inf WITHOUT clearing it first: further stacks of ONE composed graph scanned into
 * the control block the first stack's y2_grad_check cleared -- the step is then all-or-nothing over the whole graph */
int y2_grad_check_more(y2_ctx* ctx, void* ctrl, void* stream);
int y2_grad_check_full(const float* grads, size_t n, void* ctrl, void* stream);
/* Round 6.  ctrl.found_inf |= any x[i] that is non-finite or beyond +-limit; the flag is NOT cleared first (call it between
 * y2_grad_check_full and the guarded update).  For operands that a fused update rounds to a narrower type without ever
 * storing the gradient: y2_fc_adam_apply_guarded rounds dz to f16 / bf16 inside the kernel, so a |dz| above 65504 is finite in
 * every stored fp32 gradient it feeds and inf in the weight's (src/pascal/pascal_train_resnet.py:41-50, yolo_fc1). */
int y2_range_check(const float* x, size_t n, float limit, void* ctrl, void* stream);
int y2_adam_step_guarded(float* params, float* m, float* v, const float* grads, size_t n, void* ctrl, float lr,
                         float beta1, float beta2, float eps, float grad_mult, void* stream);
int y2_momentum_step_guarded(float* params, float* accum, const float* grads, size_t n, void* ctrl, float lr,
                             float momentum, float grad_mult, void* stream);

/* The optimizer step of a context's bound parameters / gradients FUSED with the re-pack of its filters into the
 * MFMA operand layouts (the update reads and writes every filter anyway; y2_forward then finds the packed copies
 * current and skips its own re-pack pass).  Same arithmetic as y2_adam_step / y2_momentum_step.  ctrl: NULL for
 * the plain step number `step`, or the guard words of the *_guarded forms (run y2_grad_check first).  Other
 * contexts bound to the same parameter buffer must still call y2_params_changed. */
/* (ctrl with step < 0: do not advance the control block -- a further stack of a composed graph whose first stack's call
 * advanced it already: ONE step counter and lr_t for the whole graph) */
int y2_adam_step_packed(y2_ctx* ctx, float* m, float* v, void* ctrl, int step, float lr, float beta1, float beta2,
                        float eps, float grad_mult, void* stream);
int y2_momentum_step_packed(y2_ctx* ctx, float* accum, void* ctrl, float lr, float momentum, float grad_mult,
                            void* stream);

/* The reference's train_op as ONE call (optimizer.minimize(loss) = compute_gradients + apply_gradients,
 * src/pascal/pascal_train_darknet.py:49-51; imagenet_train_darknet.py:58): y2_backward over all layers, then, with
 * ctrl, the sentinel overflow check (y2_grad_check), then y2_adam_step_packed / y2_momentum_step_packed -- and the
 * same results, bit for bit.  What changes is the schedule on the single-GPU path: for the pooled 3-channel first
 * layer every layer above it is checked and updated on the weight-gradient stream while the first layer's
 * gradient kernel runs (their gradients are final by then; the first layer's overflow sentinel is replaced by a
 * non-finite marker on the gradient tensor that feeds it, which decides the step before that kernel ends).
 * Data-parallel callers, whose all-reduce sits between the two halves, keep the separate calls. */
int y2_backward_adam(y2_ctx* ctx, const float* dout, float* m, float* v, void* ctrl, int step, float lr, float beta1,
                     float beta2, float eps, float grad_mult, void* stream);
int y2_backward_momentum(y2_ctx* ctx, const float* dout, float* accum, void* ctrl, float lr, float momentum,
                         float grad_mult, void* stream);

/* ---- single-op entry points (tf.nn.conv2d 'SAME' stride 1, darknet.py:20-21) used by the
 *      per-op parity tests; channel counts are padded internally to the kernels' granularity */
size_t y2_conv2d_workspace_bytes(int N, int H, int W, int Cin, int Cout, int k, int dtype);
int y2_conv2d(const float* x, const float* w, const float* bias, float* y, int N, int H, int W, int Cin,
              int Cout, int k, int dtype, void* workspace, void* stream);
int y2_conv2d_backward(const float* x, const float* w, const float* dy, float* dx, float* dw, int N, int H,
                       int W, int Cin, int Cout, int k, int dtype, void* workspace, void* stream);

/* ---- host utility: CRC-32C (Castagnoli) of a host buffer, continuing from `crc` (0 to start).  The checksum of
 *      TensorFlow's V2 checkpoint files (tensor bundle + table blocks), which the reference reads and writes through
 *      tf.train.Saver (src/yolo2_nets/net_utils.py:64-110); used by utils/tf_bundle.py on 100-MB tensors. */
uint32_t y2_crc32c(const void* data, size_t n, uint32_t crc);

#pragma GCC visibility pop
#ifdef __cplusplus
}
#endif
#endif
