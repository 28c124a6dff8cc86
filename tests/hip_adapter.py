"""Adapter exposing the HIP product path (through class Matrix -> C ABI) with the oracle's numpy
API, so one case list (tests/golden_cases.py) drives oracle, reference and GPU alike."""
import numpy as np

from convnet_amd.matrix import Matrix, make_conv_desc


def _mat(arr, rows, cols, shape4=None):
    m = Matrix()
    m.AllocateGPUMemory(rows, cols)
    m.FromNumpy(arr)
    if shape4:
        m.SetShape4D(*shape4)
    return m


def _desc(g, pool=False):
    return make_conv_desc(g.C, g.C if pool else g.F, g.Ky, g.Kx, g.sy, g.sx, g.pady, g.padx)


class HipImpl:
    kind = "hip"

    def __init__(self, fused=False):
        self.fused = fused

    # activations arrive as (C,H,W,N) numpy == column-major (N, W*H*C)
    def _act(self, a, N, W, H, C):
        return _mat(a, N, W * H * C, (N, W, H, C))

    def conv_up(self, g, images, filters, targets=None, scale_targets=0.0, scale_output=1.0):
        assert scale_output == 1.0
        x = self._act(images, g.N, g.W, g.H, g.C)
        w = _mat(filters, g.F, g.K, (g.F, g.Kx, g.Ky, g.C))
        t = self._act(targets if targets is not None else np.zeros(g.out_shape(), np.float32), g.N, g.Mx, g.My, g.F)
        Matrix.ConvUp(x, w, t, _desc(g), scale_targets)
        return t.ToNumpy().reshape(g.out_shape())

    def conv_up_bias_relu(self, g, images, filters, bias, relu=True):
        x = self._act(images, g.N, g.W, g.H, g.C)
        w = _mat(filters, g.F, g.K, (g.F, g.Kx, g.Ky, g.C))
        b = _mat(bias, 1, g.F)
        t = self._act(np.zeros(g.out_shape(), np.float32), g.N, g.Mx, g.My, g.F)
        Matrix.ConvUpBiasAct(x, w, b, t, _desc(g), 0.0, relu)
        return t.ToNumpy().reshape(g.out_shape())

    def conv_down(self, g, derivs, filters, targets=None, scale_targets=0.0, scale_output=1.0):
        assert scale_output == 1.0
        dy = self._act(derivs, g.N, g.Mx, g.My, g.F)
        w = _mat(filters, g.F, g.K, (g.F, g.Kx, g.Ky, g.C))
        t = self._act(targets if targets is not None else np.zeros(g.in_shape(), np.float32), g.N, g.W, g.H, g.C)
        Matrix.ConvDown(dy, w, t, _desc(g), scale_targets)
        return t.ToNumpy().reshape(g.in_shape())

    def conv_outp(self, g, images, derivs, targets=None, scale_targets=0.0, scale_output=1.0):
        x = self._act(images, g.N, g.W, g.H, g.C)
        dy = self._act(derivs, g.N, g.Mx, g.My, g.F)
        t = _mat(targets if targets is not None else np.zeros(g.filt_shape(), np.float32), g.F, g.K, (g.F, g.Kx, g.Ky, g.C))
        Matrix.ConvOutp(x, dy, t, _desc(g), 0, 0, scale_targets, scale_output)
        return t.ToNumpy().reshape(g.filt_shape())

    def max_pool(self, g, images, targets=None, scale_targets=0.0, scale_output=1.0):
        x = self._act(images, g.N, g.W, g.H, g.C)
        t = self._act(np.zeros(g.pooled_shape(), np.float32), g.N, g.Mx, g.My, g.C)
        Matrix.ConvMaxPool(x, t, _desc(g, True))
        return t.ToNumpy().reshape(g.pooled_shape())

    def avg_pool(self, g, images, targets=None, scale_targets=0.0, scale_output=1.0):
        x = self._act(images, g.N, g.W, g.H, g.C)
        t = self._act(np.zeros(g.pooled_shape(), np.float32), g.N, g.Mx, g.My, g.C)
        Matrix.ConvAvgPool(x, t, _desc(g, True))
        return t.ToNumpy().reshape(g.pooled_shape())

    def max_pool_undo(self, g, images, max_grads, max_acts, targets=None, scale_targets=0.0):
        x = self._act(images, g.N, g.W, g.H, g.C)
        dy = self._act(max_grads, g.N, g.Mx, g.My, g.C)
        y = self._act(max_acts, g.N, g.Mx, g.My, g.C)
        t = self._act(targets if targets is not None else np.zeros(g.in_shape(), np.float32), g.N, g.W, g.H, g.C)
        Matrix.ConvMaxPoolUndo(x, dy, y, t, _desc(g, True), scale_targets)
        return t.ToNumpy().reshape(g.in_shape())

    def avg_pool_undo(self, g, avg_grads, targets=None, scale_targets=0.0):
        dy = self._act(avg_grads, g.N, g.Mx, g.My, g.C)
        t = self._act(targets if targets is not None else np.zeros(g.in_shape(), np.float32), g.N, g.W, g.H, g.C)
        Matrix.ConvAvgPoolUndo(dy, t, _desc(g, True), scale_targets)
        return t.ToNumpy().reshape(g.in_shape())

    def rnorm(self, images, size_f, add_scale, pow_scale, blocked=False):
        C, N = images.shape[0], images.shape[-1]
        x = _mat(images, N, images.size // N)
        t = _mat(np.zeros_like(images), N, images.size // N)
        Matrix.ConvResponseNormCrossMap(x, t, C, size_f, add_scale, pow_scale, blocked)
        return t.ToNumpy().reshape(images.shape)

    def rnorm_undo(self, out_grads, inputs, size_f, add_scale, pow_scale, blocked=False):
        C, N = inputs.shape[0], inputs.shape[-1]
        dy = _mat(out_grads, N, inputs.size // N)
        x = _mat(inputs, N, inputs.size // N)
        t = _mat(np.zeros_like(inputs), N, inputs.size // N)
        Matrix.ConvResponseNormCrossMapUndo(dy, x, x, t, C, size_f, add_scale, pow_scale, blocked)
        return t.ToNumpy().reshape(inputs.shape)

    def dot(self, a, b, target, beta, alpha, a_trans=False, b_trans=False):
        A = _mat(a, a.shape[1], a.shape[0])
        B = _mat(b, b.shape[1], b.shape[0])
        T = _mat(target, target.shape[1], target.shape[0])
        Matrix.Dot(A, B, T, beta, alpha, a_trans, b_trans)   # (c, alpha=scale of c, beta=scale of product)
        return T.ToNumpy().reshape(target.shape)

    # ---- input staging (numpy (cols, rows) <-> column-major Matrix) ----------------------------------------------
    def extract_patches(self, images, wo, ho, flip, img_w, img_h, pw, ph):
        n, dims = images.shape
        colors = dims // (img_w * img_h)
        src = _mat(images, dims, n)
        dst = Matrix()
        dst.AllocateGPUMemory(n, colors * ph * pw)
        Matrix.ExtractPatches(src, dst, _mat(wo, 1, n), _mat(ho, 1, n), _mat(flip, 1, n), img_h, img_w, ph, pw)
        return dst.ToNumpy().reshape(colors, ph, pw, n)

    def shuffle_columns(self, mat, perm):
        M = _mat(mat, mat.shape[1], mat.shape[0])
        M.ShuffleColumns(_mat(perm, 1, perm.size))
        return M.ToNumpy().reshape(mat.shape)

    def add_col_mult(self, mat, vec, mult):
        M = _mat(mat, mat.shape[1], mat.shape[0])
        M.AddColVec(_mat(vec, vec.size, 1), mult)
        return M.ToNumpy().reshape(mat.shape)

    def div_by_col_vec(self, mat, vec):
        M = _mat(mat, mat.shape[1], mat.shape[0])
        M.DivideByColVec(_mat(vec, vec.size, 1))
        return M.ToNumpy().reshape(mat.shape)

    def mult_by_row_vec(self, mat, vec):
        M = _mat(mat, mat.shape[1], mat.shape[0])
        M.MultByRowVec(_mat(vec, 1, vec.size))
        return M.ToNumpy().reshape(mat.shape)

    def normalize_columns(self, mat):
        M = _mat(mat, mat.shape[1], mat.shape[0])
        M.NormalizeColumnwise()
        return M.ToNumpy().reshape(mat.shape)

    def add_to_each_pixel(self, mat1, mat2, mult):
        M = _mat(mat1, mat1.shape[1], mat1.shape[0])
        M.AddToEachPixel(_mat(mat2, mat2.shape[1], mat2.shape[0]), mult)
        return M.ToNumpy().reshape(mat1.shape)

    def copy_transpose(self, src):
        S = _mat(src, src.shape[1], src.shape[0])
        D = Matrix()
        D.AllocateGPUMemory(src.shape[0], src.shape[1])
        S.CopyTranspose(D)
        return D.ToNumpy().reshape(src.shape[1], src.shape[0])

    def add_row_vec(self, mat, vec):
        M, V = _mat(mat, mat.shape[1], mat.shape[0]), _mat(vec, 1, vec.size)
        M.AddRowVec(V)
        return M.ToNumpy().reshape(mat.shape)

    def sum_by_axis(self, mat, target, axis, mult, p):
        M = _mat(mat, mat.shape[1], mat.shape[0])
        T = _mat(target, 1, target.size) if axis == 0 else _mat(target, target.size, 1)
        (M.SumRows if axis == 0 else M.SumCols)(T, p, mult)
        return T.ToNumpy().reshape(target.shape)

    def lower_bound(self, mat, val):
        M = _mat(mat, mat.size, 1)
        M.LowerBound(val)
        return M.ToNumpy().reshape(mat.shape)

    def upper_bound_mod(self, mat, val):
        M = _mat(mat, mat.size, 1)
        M.UpperBoundMod(val)
        return M.ToNumpy().reshape(mat.shape)

    def relu_deriv(self, deriv, state):
        D, S = _mat(deriv, deriv.size, 1), _mat(state, state.size, 1)
        D.ApplyDerivativeOfReLU(S)
        return D.ToNumpy().reshape(deriv.shape)

    def softmax_row_major(self, mat):
        M = _mat(mat, mat.shape[1], mat.shape[0])
        M.ApplySoftmax()
        return M.ToNumpy().reshape(mat.shape)

    def softmax_grad_row_major(self, mat, labels):
        M, L = _mat(mat, mat.shape[1], mat.shape[0]), _mat(labels, labels.size, 1)
        T = _mat(np.zeros_like(mat), mat.shape[1], mat.shape[0])
        Matrix.SoftmaxCEDeriv(M, L, T)
        return T.ToNumpy().reshape(mat.shape)

    def softmax_correct_row_major(self, mat, labels):
        M, L = _mat(mat, mat.shape[1], mat.shape[0]), _mat(labels, labels.size, 1)
        T = _mat(np.zeros(labels.size, np.float32), labels.size, 1)
        Matrix.SoftmaxCorrect(M, L, T)
        return T.ToNumpy().reshape(-1)

    def softmax_ce_row_major(self, mat, labels, tiny=1e-10):
        M, L = _mat(mat, mat.shape[1], mat.shape[0]), _mat(labels, labels.size, 1)
        T = _mat(np.zeros(labels.size, np.float32), labels.size, 1)
        Matrix.SoftmaxCE(M, L, T)
        return T.ToNumpy().reshape(-1)

    def normlimit_rows(self, mat, norm, constraint):
        M = _mat(mat, mat.shape[1], mat.shape[0])
        M.NormLimitByAxis(1, norm, constraint)
        return M.ToNumpy().reshape(mat.shape)

    def sgd_step(self, grad, param, history, l2_decay, gradient_clip, epsilon, momentum, norm_limit=0.0, norm_constraint=0.0):
        G, W, H = (_mat(a, a.shape[1], a.shape[0]) for a in (grad, param, history))
        if self.fused:
            Matrix.SGDMomentumStep(G, W, H, l2_decay, gradient_clip, epsilon, momentum)
        else:  # the reference's own op sequence (src/optimizer.cc:174-200)
            if l2_decay > 0:
                G.Add(W, l2_decay)
            if gradient_clip > 0:
                G.UpperBoundMod(gradient_clip)
            G.Mult(epsilon)
            H.Mult(momentum)
            H.Add(G)
            W.Add(H, -1)
        if norm_constraint > 0:
            W.NormLimitByAxis(1, norm_constraint, True)
        elif norm_limit > 0:
            W.NormLimitByAxis(1, norm_limit, False)
        grad[...] = G.ToNumpy().reshape(grad.shape)
        param[...] = W.ToNumpy().reshape(param.shape)
        history[...] = H.ToNumpy().reshape(history.shape)


def conv_outp_bias(g, images, derivs, dw0, db0, scale_targets, scale_output):
    """convOutpBias through the ABI: returns (dW, db)."""
    h = HipImpl()
    x = h._act(images, g.N, g.W, g.H, g.C)
    dy = h._act(derivs, g.N, g.Mx, g.My, g.F)
    t = _mat(dw0, g.F, g.K, (g.F, g.Kx, g.Ky, g.C))
    b = _mat(db0.reshape(1, g.F), 1, g.F)
    Matrix.ConvOutpBias(x, dy, t, b, _desc(g), scale_targets, scale_output)
    return t.ToNumpy().reshape(g.filt_shape()), b.ToNumpy().reshape(g.F)
