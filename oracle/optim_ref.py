"""numpy restatement of the two optimizers the reference's train scripts use.
TEST INFRASTRUCTURE ONLY.

  * tf.train.AdamOptimizer() defaults (src/pascal/pascal_train_darknet.py:51):
    lr 1e-3, beta1 0.9, beta2 0.999, eps 1e-8, TF's "epsilon-hat" form
        lr_t = lr * sqrt(1 - beta2^t) / (1 - beta1^t)
        m <- beta1*m + (1-beta1)*g ; v <- beta2*v + (1-beta2)*g*g
        var <- var - lr_t * m / (sqrt(v) + eps)
  * tf.train.MomentumOptimizer(0.001, 0.9) (src/imagenet/imagenet_train_darknet.py:58):
        accum <- momentum*accum + g ; var <- var - lr*accum      (no Nesterov)
"""
import numpy as np


def adam_step(var, m, v, g, t, lr=1e-3, beta1=0.9, beta2=0.999, eps=1e-8, dtype=np.float32):
    var, m, v, g = (a.astype(dtype) for a in (var, m, v, g))
    lr_t = dtype(lr) * np.sqrt(dtype(1.0) - dtype(beta2) ** dtype(t)) / (dtype(1.0) - dtype(beta1) ** dtype(t))
    m = dtype(beta1) * m + dtype(1.0 - beta1) * g
    v = dtype(beta2) * v + dtype(1.0 - beta2) * g * g
    var = var - lr_t * m / (np.sqrt(v) + dtype(eps))
    return var, m, v


def momentum_step(var, accum, g, lr=1e-3, momentum=0.9, dtype=np.float32):
    var, accum, g = (a.astype(dtype) for a in (var, accum, g))
    accum = dtype(momentum) * accum + g
    var = var - dtype(lr) * accum
    return var, accum
