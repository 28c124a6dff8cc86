"""TEST INFRASTRUCTURE — NOT PRODUCT CODE.

numpy/ctypes front-end for the two CPU checkers of the hot path:

* ``port``  — ``oracle/convnet_oracle.c`` (``liboracle.so``): the portable plain-C restatement of
  the reference algorithms, each function citing the reference file:line it follows.
* ``ref``   — ``oracle/_ref/libconvnet_ref.so``: the reference's OWN CPU code
  (eigenmat/*.cc + src/CPUMatrix.cc) compiled unmodified by ``oracle/Makefile`` (only available
  where it has been built; ``None`` otherwise).

Only ``tests/``, ``__graft_entry__.smoke()`` and ``bench.py``'s ``cpu_baseline`` leg may import
this package.  ``convnet_amd`` (the product) never does.

All arrays are float32 and use the reference memory layout: an activation of logical shape
(N, C, H, W) is passed as a numpy array of shape ``(C, H, W, N)`` (C-contiguous), whose bytes are
exactly the reference's column-major ``(N, X*Y*C)`` matrix; filters are ``(C, Ky, Kx, F)``.
"""
import ctypes
import os
import subprocess
from dataclasses import dataclass

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_f32p = ctypes.POINTER(ctypes.c_float)
_i32p = ctypes.POINTER(ctypes.c_int)


def build(verbose=False):
    """Compile liboracle.so (always) and oracle/_ref (when /root/reference is mounted)."""
    out = subprocess.run(["make", "-C", _HERE, "all"], capture_output=True, text=True)
    if verbose or out.returncode != 0:
        print(out.stdout, out.stderr)
    if out.returncode != 0:
        raise RuntimeError("oracle build failed")


def _p(a):
    assert a.dtype == np.float32 and a.flags["C_CONTIGUOUS"], (a.dtype, a.flags)
    return a.ctypes.data_as(_f32p)


@dataclass(frozen=True)
class Geom:
    """One conv/pool geometry.  ``pad`` is the pbtxt (positive) padding; the reference's ConvDesc
    stores ``-pad`` (src/edge.cc:97-99) and that negation is applied here."""
    N: int
    C: int
    H: int
    W: int
    F: int
    Ky: int
    Kx: int
    sy: int = 1
    sx: int = 1
    pady: int = 0
    padx: int = 0

    @property
    def My(self):  # src/edge.cc:111
        return (self.H + 2 * self.pady - self.Ky) // self.sy + 1

    @property
    def Mx(self):
        return (self.W + 2 * self.padx - self.Kx) // self.sx + 1

    @property
    def K(self):
        return self.C * self.Ky * self.Kx

    def desc8(self):
        return (ctypes.c_int * 8)(self.C, self.F, self.Ky, self.Kx, self.sy, self.sx, -self.pady, -self.padx)

    def in_shape(self):
        return (self.C, self.H, self.W, self.N)

    def out_shape(self):
        return (self.F, self.My, self.Mx, self.N)

    def filt_shape(self):
        return (self.C, self.Ky, self.Kx, self.F)

    def pooled_shape(self):
        return (self.C, self.My, self.Mx, self.N)


class _Port:
    kind = "port"

    def __init__(self, path):
        self.lib = ctypes.CDLL(path)
        self.lib.oracle_version.restype = ctypes.c_int

    # --- conv -------------------------------------------------------------------------------
    def _conv_args(self, g):
        ci = ctypes.c_int
        return [ci(g.N), ci(g.C), ci(g.H), ci(g.W), ci(g.F), ci(g.Ky), ci(g.Kx), ci(g.sy), ci(g.sx),
                ci(-g.pady), ci(-g.padx), ci(g.My), ci(g.Mx)]

    def conv_up(self, g, images, filters, targets=None, scale_targets=0.0, scale_output=1.0):
        t = np.zeros(g.out_shape(), np.float32) if targets is None else targets
        self.lib.oracle_conv_up(_p(images), _p(filters), _p(t), *self._conv_args(g),
                                ctypes.c_float(scale_targets), ctypes.c_float(scale_output))
        return t

    def conv_down(self, g, derivs, filters, targets=None, scale_targets=0.0, scale_output=1.0):
        t = np.zeros(g.in_shape(), np.float32) if targets is None else targets
        self.lib.oracle_conv_down(_p(derivs), _p(filters), _p(t), *self._conv_args(g),
                                  ctypes.c_float(scale_targets), ctypes.c_float(scale_output))
        return t

    def conv_outp(self, g, images, derivs, targets=None, scale_targets=0.0, scale_output=1.0):
        t = np.zeros(g.filt_shape(), np.float32) if targets is None else targets
        self.lib.oracle_conv_outp(_p(images), _p(derivs), _p(t), *self._conv_args(g),
                                  ctypes.c_float(scale_targets), ctypes.c_float(scale_output))
        return t

    # --- pool -------------------------------------------------------------------------------
    def _pool_args(self, g):
        ci = ctypes.c_int
        return [ci(g.N), ci(g.C), ci(g.H), ci(g.W), ci(g.Ky), ci(g.Kx), ci(g.sy), ci(g.sx),
                ci(-g.pady), ci(-g.padx), ci(g.My), ci(g.Mx)]

    def max_pool(self, g, images, targets=None, scale_targets=0.0, scale_output=1.0):
        t = np.zeros(g.pooled_shape(), np.float32) if targets is None else targets
        self.lib.oracle_max_pool(_p(images), _p(t), *self._pool_args(g),
                                 ctypes.c_float(scale_targets), ctypes.c_float(scale_output))
        return t

    def avg_pool(self, g, images, targets=None, scale_targets=0.0, scale_output=1.0):
        t = np.zeros(g.pooled_shape(), np.float32) if targets is None else targets
        self.lib.oracle_avg_pool(_p(images), _p(t), *self._pool_args(g),
                                 ctypes.c_float(scale_targets), ctypes.c_float(scale_output))
        return t

    def max_pool_undo(self, g, images, max_grads, max_acts, targets=None, scale_targets=0.0):
        t = np.zeros(g.in_shape(), np.float32) if targets is None else targets
        self.lib.oracle_max_pool_undo(_p(images), _p(max_grads), _p(max_acts), _p(t),
                                      *self._pool_args(g), ctypes.c_float(scale_targets))
        return t

    def avg_pool_undo(self, g, avg_grads, targets=None, scale_targets=0.0):
        t = np.zeros(g.in_shape(), np.float32) if targets is None else targets
        self.lib.oracle_avg_pool_undo(_p(avg_grads), _p(t), *self._pool_args(g),
                                      ctypes.c_float(scale_targets))
        return t

    # --- response norm (arrays (C, H, W, N)) ---------------------------------------------------
    def rnorm(self, images, size_f, add_scale, pow_scale, blocked=False):
        C = images.shape[0]
        t = np.zeros_like(images)
        self.lib.oracle_rnorm(_p(images), _p(t), ctypes.c_int(images.size // C), ctypes.c_int(C),
                              ctypes.c_int(size_f), ctypes.c_float(add_scale),
                              ctypes.c_float(pow_scale), ctypes.c_int(int(blocked)))
        return t

    def rnorm_undo(self, out_grads, inputs, size_f, add_scale, pow_scale, blocked=False):
        C = inputs.shape[0]
        t = np.zeros_like(inputs)
        self.lib.oracle_rnorm_undo(_p(out_grads), _p(inputs), _p(t), ctypes.c_int(inputs.size // C),
                                   ctypes.c_int(C), ctypes.c_int(size_f), ctypes.c_float(add_scale),
                                   ctypes.c_float(pow_scale), ctypes.c_int(int(blocked)))
        return t

    # --- dense: matrices are passed as numpy arrays of shape (cols, rows) = column-major bytes --
    def dot(self, a, b, target, beta, alpha, a_trans=False, b_trans=False):
        """target = beta*target + alpha*op(a)@op(b).  ``a`` etc. are (cols, rows)-shaped numpy views of
        column-major (rows, cols) matrices."""
        self.lib.oracle_dot(_p(a), ctypes.c_int(a.shape[1]), ctypes.c_int(a.shape[0]), ctypes.c_int(int(a_trans)),
                            _p(b), ctypes.c_int(b.shape[1]), ctypes.c_int(b.shape[0]), ctypes.c_int(int(b_trans)),
                            _p(target), ctypes.c_float(beta), ctypes.c_float(alpha))
        return target

    def add_row_vec(self, mat, vec):
        self.lib.oracle_add_row_vec(_p(mat), _p(vec), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]))
        return mat

    def sum_by_axis(self, mat, target, axis, mult, p):
        self.lib.oracle_sum_by_axis(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]),
                                    _p(target), ctypes.c_int(axis), ctypes.c_float(mult), ctypes.c_float(p))
        return target

    def lower_bound(self, mat, val):
        self.lib.oracle_lower_bound(_p(mat), ctypes.c_size_t(mat.size), ctypes.c_float(val))
        return mat

    def upper_bound_mod(self, mat, val):
        self.lib.oracle_upper_bound_mod(_p(mat), ctypes.c_size_t(mat.size), ctypes.c_float(val))
        return mat

    def relu_deriv(self, deriv, state):
        self.lib.oracle_relu_deriv(_p(deriv), _p(state), ctypes.c_size_t(deriv.size))
        return deriv

    def softmax_row_major(self, mat):
        self.lib.oracle_softmax_row_major(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]))
        return mat

    def softmax_grad_row_major(self, mat, labels):
        t = np.zeros_like(mat)
        self.lib.oracle_softmax_grad_row_major(_p(mat), _p(labels), _p(t), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]))
        return t

    def softmax_correct_row_major(self, mat, labels):
        t = np.zeros(mat.shape[1], np.float32)
        self.lib.oracle_softmax_correct_row_major(_p(mat), _p(labels), _p(t), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]))
        return t

    def softmax_ce_row_major(self, mat, labels, tiny=1e-10):
        t = np.zeros(mat.shape[1], np.float32)
        self.lib.oracle_softmax_ce_row_major(_p(mat), _p(labels), _p(t), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), ctypes.c_float(tiny))
        return t

    def normlimit_rows(self, mat, norm, constraint):
        self.lib.oracle_normlimit_rows(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), ctypes.c_float(norm), ctypes.c_int(int(constraint)))
        return mat

    def sgd_step(self, grad, param, history, l2_decay, gradient_clip, epsilon, momentum, norm_limit=0.0, norm_constraint=0.0):
        self.lib.oracle_sgd_step(_p(grad), _p(param), _p(history), ctypes.c_int(param.shape[1]), ctypes.c_int(param.shape[0]),
                                 ctypes.c_float(l2_decay), ctypes.c_float(gradient_clip), ctypes.c_float(epsilon),
                                 ctypes.c_float(momentum), ctypes.c_float(norm_limit), ctypes.c_float(norm_constraint))

    def dropout(self, mat, uniform, dropprob, val, scale):
        self.lib.oracle_dropout(_p(mat), _p(uniform), ctypes.c_size_t(mat.size), ctypes.c_float(dropprob), ctypes.c_float(val), ctypes.c_float(scale))
        return mat

    # ---- input staging: matrices are numpy arrays of shape (cols, rows) holding the column-major bytes -------------
    def extract_patches(self, images, wo, ho, flip, img_w, img_h, pw, ph):
        """images (num_images, dims) -> CHWN batch (colors, ph, pw, num_images)."""
        n, dims = images.shape
        colors = dims // (img_w * img_h)
        out = np.zeros((colors, ph, pw, n), np.float32)
        self._extract(images, n, dims, colors, out, wo, ho, flip, img_w, img_h, pw, ph)
        return out

    def _extract(self, images, n, dims, colors, out, wo, ho, flip, img_w, img_h, pw, ph):
        ci = ctypes.c_int
        self.lib.oracle_extract_patches(_p(images), ci(n), ci(colors), _p(out), _p(wo), _p(ho), _p(flip), ci(img_w), ci(img_h), ci(pw), ci(ph))

    def shuffle_columns(self, mat, perm):
        self.lib.oracle_shuffle_columns(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), _p(perm))
        return mat

    def add_col_mult(self, mat, vec, mult):
        self.lib.oracle_add_col_mult(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), _p(vec), ctypes.c_float(mult))
        return mat

    def div_by_col_vec(self, mat, vec):
        self.lib.oracle_div_by_col_vec(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), _p(vec))
        return mat

    def mult_by_row_vec(self, mat, vec):
        self.lib.oracle_mult_by_row_vec(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), _p(vec))
        return mat

    def normalize_columns(self, mat):
        self.lib.oracle_normalize_columns(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]))
        return mat

    def add_to_each_pixel(self, mat1, mat2, mult):
        self.lib.oracle_add_to_each_pixel(_p(mat1), ctypes.c_int(mat1.shape[1]), ctypes.c_int(mat1.shape[0]), _p(mat2),
                                          ctypes.c_int(mat2.shape[0]), ctypes.c_float(mult))
        return mat1

    def copy_transpose(self, src):
        dst = np.zeros((src.shape[1], src.shape[0]), np.float32)
        self.lib.oracle_copy_transpose(_p(src), ctypes.c_int(src.shape[1]), ctypes.c_int(src.shape[0]), _p(dst))
        return dst


class _Ref(_Port):
    """Same numpy API, executed by the reference's own compiled code."""
    kind = "reference"

    def __init__(self, path):
        self.lib = ctypes.CDLL(path)

    def _g(self, g):
        ci = ctypes.c_int
        return [ci(g.N), ci(g.H), ci(g.W), ci(g.My), ci(g.Mx), g.desc8()]

    def conv_up(self, g, images, filters, targets=None, scale_targets=0.0, scale_output=1.0):
        t = np.zeros(g.out_shape(), np.float32) if targets is None else targets
        self.lib.ref_conv_up(_p(images), _p(filters), _p(t), *self._g(g), ctypes.c_float(scale_targets), ctypes.c_float(scale_output))
        return t

    def conv_down(self, g, derivs, filters, targets=None, scale_targets=0.0, scale_output=1.0):
        t = np.zeros(g.in_shape(), np.float32) if targets is None else targets
        self.lib.ref_conv_down(_p(derivs), _p(filters), _p(t), *self._g(g), ctypes.c_float(scale_targets), ctypes.c_float(scale_output))
        return t

    def conv_outp(self, g, images, derivs, targets=None, scale_targets=0.0, scale_output=1.0):
        t = np.zeros(g.filt_shape(), np.float32) if targets is None else targets
        self.lib.ref_conv_outp(_p(images), _p(derivs), _p(t), *self._g(g), ctypes.c_float(scale_targets), ctypes.c_float(scale_output))
        return t

    def max_pool(self, g, images, targets=None, scale_targets=0.0, scale_output=1.0):
        assert scale_targets == 0.0 and scale_output == 1.0  # CPU class hard-codes these
        t = np.zeros(g.pooled_shape(), np.float32)
        self.lib.ref_max_pool(_p(images), _p(t), *self._g(g))
        return t

    def avg_pool(self, g, images, targets=None, scale_targets=0.0, scale_output=1.0):
        assert scale_targets == 0.0 and scale_output == 1.0
        t = np.zeros(g.pooled_shape(), np.float32)
        self.lib.ref_avg_pool(_p(images), _p(t), *self._g(g))
        return t

    def max_pool_undo(self, g, images, max_grads, max_acts, targets=None, scale_targets=0.0):
        t = np.zeros(g.in_shape(), np.float32) if targets is None else targets
        self.lib.ref_max_pool_undo(_p(images), _p(max_grads), _p(max_acts), _p(t), *self._g(g), ctypes.c_float(scale_targets))
        return t

    def avg_pool_undo(self, g, avg_grads, targets=None, scale_targets=0.0):
        t = np.zeros(g.in_shape(), np.float32) if targets is None else targets
        self.lib.ref_avg_pool_undo(_p(avg_grads), _p(t), *self._g(g), ctypes.c_float(scale_targets))
        return t

    def rnorm(self, images, size_f, add_scale, pow_scale, blocked=False):
        C, N = images.shape[0], images.shape[-1]
        t = np.zeros_like(images)
        self.lib.ref_rnorm(_p(images), _p(t), ctypes.c_int(N), ctypes.c_int(images.size // (C * N)), ctypes.c_int(C),
                           ctypes.c_int(size_f), ctypes.c_float(add_scale), ctypes.c_float(pow_scale), ctypes.c_int(int(blocked)))
        return t

    def rnorm_undo(self, out_grads, inputs, size_f, add_scale, pow_scale, blocked=False):
        C, N = inputs.shape[0], inputs.shape[-1]
        t = np.zeros_like(inputs)
        self.lib.ref_rnorm_undo(_p(out_grads), _p(inputs), _p(t), ctypes.c_int(N), ctypes.c_int(inputs.size // (C * N)), ctypes.c_int(C),
                                ctypes.c_int(size_f), ctypes.c_float(add_scale), ctypes.c_float(pow_scale), ctypes.c_int(int(blocked)))
        return t

    def dot(self, a, b, target, beta, alpha, a_trans=False, b_trans=False):
        rc = self.lib.ref_dot(_p(a), ctypes.c_int(a.shape[1]), ctypes.c_int(a.shape[0]), ctypes.c_int(int(a_trans)),
                              _p(b), ctypes.c_int(b.shape[1]), ctypes.c_int(b.shape[0]), ctypes.c_int(int(b_trans)),
                              _p(target), ctypes.c_int(target.shape[1]), ctypes.c_int(target.shape[0]),
                              ctypes.c_float(beta), ctypes.c_float(alpha))
        assert rc == 0, rc
        return target

    def add_row_vec(self, mat, vec):
        assert self.lib.ref_add_row_vec(_p(mat), _p(vec), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0])) == 0
        return mat

    def sum_by_axis(self, mat, target, axis, mult, p):
        assert self.lib.ref_sum_by_axis(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), _p(target),
                                        ctypes.c_int(axis), ctypes.c_float(mult), ctypes.c_float(p)) == 0
        return target

    def lower_bound(self, mat, val):
        self.lib.ref_lower_bound_scalar(_p(mat), ctypes.c_int(mat.size), ctypes.c_float(val))
        return mat

    def upper_bound_mod(self, mat, val):
        self.lib.ref_upper_bound_mod_scalar(_p(mat), ctypes.c_int(mat.size), ctypes.c_float(val))
        return mat

    def relu_deriv(self, deriv, state):
        self.lib.ref_relu_deriv(_p(deriv), _p(state), ctypes.c_int(deriv.size))
        return deriv

    def softmax_row_major(self, mat):
        self.lib.ref_softmax_row_major(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]))
        return mat

    def softmax_grad_row_major(self, mat, labels):
        t = np.zeros_like(mat)
        self.lib.ref_softmax_grad_row_major(_p(mat), _p(labels), _p(t), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]))
        return t

    def softmax_correct_row_major(self, mat, labels):
        t = np.zeros(mat.shape[1], np.float32)
        self.lib.ref_softmax_correct_row_major(_p(mat), _p(labels), _p(t), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]))
        return t

    def softmax_ce_row_major(self, mat, labels, tiny=1e-10):
        t = np.zeros(mat.shape[1], np.float32)
        self.lib.ref_softmax_ce_row_major(_p(mat), _p(labels), _p(t), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), ctypes.c_float(tiny))
        return t

    def normlimit_rows(self, mat, norm, constraint):
        self.lib.ref_normlimit_by_axis(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), ctypes.c_int(1), ctypes.c_float(norm), ctypes.c_int(int(constraint)))
        return mat

    def _extract(self, images, n, dims, colors, out, wo, ho, flip, img_w, img_h, pw, ph):
        ci = ctypes.c_int
        assert self.lib.ref_extract_patches(_p(images), ci(dims), ci(n), _p(out), _p(wo), _p(ho), _p(flip), ci(img_w), ci(img_h), ci(pw), ci(ph)) == 0

    def shuffle_columns(self, mat, perm):
        assert self.lib.ref_shuffle_columns(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), _p(perm)) == 0
        return mat

    def add_col_mult(self, mat, vec, mult):
        assert self.lib.ref_add_col_mult(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), _p(vec), ctypes.c_float(mult)) == 0
        return mat

    def div_by_col_vec(self, mat, vec):
        assert self.lib.ref_div_by_col_vec(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), _p(vec)) == 0
        return mat

    def mult_by_row_vec(self, mat, vec):
        assert self.lib.ref_mult_by_row_vec(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0]), _p(vec)) == 0
        return mat

    def normalize_columns(self, mat):
        assert self.lib.ref_normalize_columns(_p(mat), ctypes.c_int(mat.shape[1]), ctypes.c_int(mat.shape[0])) == 0
        return mat

    def add_to_each_pixel(self, mat1, mat2, mult):
        assert self.lib.ref_add_to_each_pixel(_p(mat1), ctypes.c_int(mat1.shape[1]), ctypes.c_int(mat1.shape[0]), _p(mat2),
                                              ctypes.c_int(mat2.shape[0]), ctypes.c_float(mult)) == 0
        return mat1

    def copy_transpose(self, src):
        dst = np.zeros((src.shape[1], src.shape[0]), np.float32)
        assert self.lib.ref_copy_transpose(_p(src), ctypes.c_int(src.shape[1]), ctypes.c_int(src.shape[0]), _p(dst)) == 0
        return dst

    def sgd_step(self, grad, param, history, l2_decay, gradient_clip, epsilon, momentum, norm_limit=0.0, norm_constraint=0.0):
        self.lib.ref_sgd_step(_p(grad), _p(param), _p(history), ctypes.c_int(param.shape[1]), ctypes.c_int(param.shape[0]),
                              ctypes.c_float(l2_decay), ctypes.c_float(gradient_clip), ctypes.c_float(epsilon),
                              ctypes.c_float(momentum), ctypes.c_float(norm_limit), ctypes.c_float(norm_constraint))

    def dropout(self, *a, **k):
        raise NotImplementedError("reference dropout draws from its own RNG stream")


def _load():
    port_path = os.path.join(_HERE, "liboracle.so")
    if not os.path.exists(port_path):
        build()
    port = _Port(port_path)
    ref_path = os.path.join(_HERE, "_ref", "libconvnet_ref.so")
    ref = None
    if os.path.exists(ref_path):
        try:
            ref = _Ref(ref_path)
        except OSError:
            ref = None
    return port, ref


port, ref = _load()
