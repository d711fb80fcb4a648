"""oracle (TEST INFRASTRUCTURE ONLY): tf.optimizers.Adam as the reference uses it
(model/model_util/optimizers.py:7-13: `tf.optimizers.Adam(lr=basic_lr)`, i.e. beta_1 0.9, beta_2 0.999, epsilon 1e-7,
amsgrad False; applied by model/train_val.py:86 `optimizer.apply_gradients(zip(grads, weights))`).

Keras OptimizerV2 Adam (tensorflow==2.4.1, keras/optimizer_v2/adam.py `_resource_apply_dense` -> ResourceApplyAdam), the
non-amsgrad update for step t = iterations + 1:
    lr_t = lr * sqrt(1 - beta_2^t) / (1 - beta_1^t)
    m    = beta_1 m + (1 - beta_1) g
    v    = beta_2 v + (1 - beta_2) g^2
    var  = var - lr_t * m / (sqrt(v) + epsilon)          (epsilon on the UNcorrected sqrt(v): "epsilon hat")
Keras holds the hyper-parameters in the VARIABLE's dtype (float32: `_prepare_local` / `_get_hyper(name, var_dtype)`), so
beta_2 is float32(0.999) = 0.99900001287..., and 1 - beta_2 = 9.9998713e-4 (not 1e-3): the restatement rounds lr, beta_1,
beta_2 and epsilon to float32 first and then evaluates the update in float64.
Third-party arithmetic (not under the reference checkout): parity pinned only by this published update rule.
"""
import numpy as np


class KerasAdamRef:
    def __init__(self, learning_rate, beta_1=0.9, beta_2=0.999, epsilon=1e-7):
        f32 = lambda v: float(np.float32(v))                     # hyper-parameters live in float32 (the variable dtype)
        self.lr, self.b1, self.b2, self.eps = f32(learning_rate), f32(beta_1), f32(beta_2), f32(epsilon)
        self.iterations = 0
        self.m = self.v = None

    def apply_gradients(self, var, grad):
        """var, grad: float64 numpy arrays of one shape; returns the updated variable (state kept inside)."""
        var, grad = np.asarray(var, dtype=np.float64), np.asarray(grad, dtype=np.float64)
        if self.m is None:
            self.m, self.v = np.zeros_like(var), np.zeros_like(var)
        self.iterations += 1
        t = self.iterations
        lr_t = self.lr * np.sqrt(1.0 - self.b2 ** t) / (1.0 - self.b1 ** t)
        self.m = self.b1 * self.m + (1.0 - self.b1) * grad
        self.v = self.b2 * self.v + (1.0 - self.b2) * grad * grad
        return var - lr_t * self.m / (np.sqrt(self.v) + self.eps)
