"""Seeded case list shared by tests/golden/make_golden.py (reference -> fixture), the CPU oracle
tests and the GPU parity tests.  ``compute_all(impl)`` replays every case on any object exposing
the oracle numpy API (oracle.port, oracle.ref, or tests/hip_adapter.HipImpl)."""
import numpy as np

from oracle import Geom

CONV = {
    "test2d_small": Geom(N=4, C=32, H=12, W=12, F=64, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),
    "ragged": Geom(N=5, C=3, H=17, W=15, F=7, Ky=5, Kx=3, sy=2, sx=1, pady=2, padx=1),
    "mnist_conv1": Geom(N=2, C=1, H=28, W=28, F=48, Ky=4, Kx=4),
    "alex_conv3_tiny": Geom(N=2, C=16, H=13, W=13, F=24, Ky=3, Kx=3, pady=1, padx=1),
}
POOL = {
    "alex_pool": Geom(N=7, C=5, H=11, W=11, F=5, Ky=3, Kx=3, sy=2, sx=2, pady=1, padx=1),
    "mnist_pool": Geom(N=6, C=4, H=25, W=25, F=4, Ky=4, Kx=4, sy=2, sx=2),
}
RNORM = {"rn_a": ((32, 6, 6, 8), 8, False), "rn_b": ((96, 3, 3, 5), 24, False), "rn_blk": ((20, 2, 3, 4), 5, True)}


def inputs(seed, *shapes):
    rng = np.random.default_rng(seed)
    return [rng.standard_normal(s).astype(np.float32) for s in shapes]


def compute_all(R):
    out = {}
    for i, (name, g) in enumerate(CONV.items()):
        x, w, dy = inputs(100 + i, g.in_shape(), g.filt_shape(), g.out_shape())
        out[f"conv/{name}/up"] = R.conv_up(g, x, w)
        out[f"conv/{name}/down"] = R.conv_down(g, dy, w)
        out[f"conv/{name}/outp"] = R.conv_outp(g, x, dy, None, 0.0, 1.0 / g.N)
    for i, (name, g) in enumerate(POOL.items()):
        x, dy = inputs(200 + i, g.in_shape(), g.pooled_shape())
        x = np.maximum(x, 0)
        mp = R.max_pool(g, x)
        out[f"pool/{name}/max"] = mp
        out[f"pool/{name}/avg"] = R.avg_pool(g, x)
        out[f"pool/{name}/max_undo"] = R.max_pool_undo(g, x, dy, mp)
        out[f"pool/{name}/avg_undo"] = R.avg_pool_undo(g, dy)
    for i, (name, (shape, size_f, blocked)) in enumerate(RNORM.items()):
        x, dy = inputs(300 + i, shape, shape)
        out[f"rnorm/{name}/fwd"] = R.rnorm(x, size_f, 0.005, 0.75, blocked)
        out[f"rnorm/{name}/undo"] = R.rnorm_undo(dy, x, size_f, 0.005, 0.75, blocked)
    x, w, dy = inputs(400, (37, 9), (37, 11), (11, 9))
    out["fc/up"] = R.dot(x, w, np.zeros((11, 9), np.float32), 0.0, 1.0, False, True)
    out["fc/down"] = R.dot(dy, w, np.zeros((37, 9), np.float32), 0.0, 1.0)
    out["fc/outp"] = R.dot(dy, x, np.zeros((37, 11), np.float32), 0.0, 1.0 / 9, True, False)
    (z,) = inputs(401, (10, 13))
    labels = np.random.default_rng(402).integers(0, 10, 13).astype(np.float32)
    p = R.softmax_row_major(3 * z)
    out["softmax/p"] = p
    out["softmax/grad"] = R.softmax_grad_row_major(p, labels)
    out["softmax/correct"] = R.softmax_correct_row_major(p, labels)
    out["softmax/ce"] = R.softmax_ce_row_major(p, labels)
    g0, w0, h0 = inputs(403, (30, 12), (30, 12), (30, 12))
    R.sgd_step(g0, w0, h0, 5e-4, 0.9, 0.01, 0.7, 0.8, 0.0)
    out["sgd/grad"], out["sgd/param"], out["sgd/hist"] = g0, w0, h0
    out.update(staging(R))
    return out


def staging_inputs():
    """The DataHandler's GPU-side staging ops: 11 cases of 3 x 20 x 23 images, 13 x 16 patches."""
    rng = np.random.default_rng(500)
    n, colors, W, H, pw, ph = 11, 3, 23, 20, 16, 13
    return dict(n=n, colors=colors, W=W, H=H, pw=pw, ph=ph,
                images=rng.standard_normal((n, colors * H * W)).astype(np.float32),
                wo=rng.integers(0, W - pw + 1, n).astype(np.float32), ho=rng.integers(0, H - ph + 1, n).astype(np.float32),
                flip=(rng.random(n) > 0.5).astype(np.float32), perm=rng.permutation(n).astype(np.float32),
                mean=rng.standard_normal(colors * H * W).astype(np.float32),
                std=(rng.random(colors * H * W) + 0.5).astype(np.float32),
                noise=rng.standard_normal((colors, n)).astype(np.float32), rowvec=rng.standard_normal(colors).astype(np.float32))


def staging(R):
    d = staging_inputs()
    out = {}
    out["staging/extract"] = R.extract_patches(d["images"], d["wo"], d["ho"], d["flip"], d["W"], d["H"], d["pw"], d["ph"])
    out["staging/shuffle"] = R.shuffle_columns(d["images"].copy(), d["perm"])
    m = R.add_col_mult(d["images"].copy(), d["mean"], -1.0)
    out["staging/normalize"] = R.div_by_col_vec(m, d["std"])
    out["staging/center"] = R.normalize_columns(d["images"].copy())
    out["staging/transpose"] = R.copy_transpose(d["images"])
    batch = out["staging/extract"].reshape(d["colors"] * d["ph"] * d["pw"], d["n"]).copy()    # (cols, rows) view of the CHWN batch
    out["staging/pixel_noise"] = R.add_to_each_pixel(batch, d["noise"], 0.1)
    out["staging/mult_row"] = R.mult_by_row_vec(d["noise"].copy(), d["rowvec"])
    return out


def rel_err(a, b):
    """The reference's own acceptance metric: max|a-b| / mean|a+b| (py/test_conv.py:382-385)."""
    a = np.asarray(a, np.float64)
    b = np.asarray(b, np.float64)
    return float(np.abs(a - b).max() / (np.abs(a + b).mean() + 1e-30))
