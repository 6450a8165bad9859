// The two optimizer updates as ONE piece of device code shared by the flat kernels (optim.hip) and the fused
// update + re-pack kernel (pack.hip): with contraction pinned, both produce the same bits.
//   tf.train.AdamOptimizer (epsilon-hat form, src/pascal/pascal_train_darknet.py:51):
//       m <- b1 m + (1 - b1) g ;  v <- b2 v + (1 - b2) g g ;  var <- var - lr_t m / (sqrt(v) + eps)
//   tf.train.MomentumOptimizer (src/imagenet/imagenet_train_darknet.py:58):
//       accum <- momentum accum + g ;  var <- var - lr accum
#pragma once
#include "common.h"

namespace y2 {

Y2_DEV void adam_update(float& p, float& m, float& v, float g, float lr_t, float b1, float b2, float eps) {
#pragma clang fp contract(off)
    const float gm = (1.0f - b1) * g;
    const float gv = ((1.0f - b2) * g) * g;
    m = __builtin_fmaf(b1, m, gm);
    v = __builtin_fmaf(b2, v, gv);
    const float step = (lr_t * m) / (sqrtf(v) + eps);
    p = p - step;
}

Y2_DEV void momentum_update(float& p, float& acc, float g, float lr, float mom) {
#pragma clang fp contract(off)
    acc = __builtin_fmaf(mom, acc, g);
    const float step = lr * acc;
    p = p - step;
}

}  // namespace y2
