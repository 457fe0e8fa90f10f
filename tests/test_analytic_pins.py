"""Closed-form pins that do NOT come from the shader text: a homogeneous medium has textbook answers.

The oracle and the kernels are restatements of the reference's GLSL by one author (parity unpinned).
These tests check both against physics written down independently, in float64 NumPy, from nothing but the scene:

  * deterministic DVR: front-to-back compositing of a constant extinction sigma over a path of length l is
    (1 - exp(-sigma l)) * colour * K (Beer-Lambert), K the single-scatter gain of fragment.frag:94-97;
  * the three stochastic modes at one bounce with the directional light and no environment term estimate the
    single-scatter integral  albedo * colour * f_p * Le * Int sigma exp(-sigma (t - t_near)) exp(-sigma s(t)) dt,
    s(t) the distance from the collision point to the box exit towards the light.  `default` and `no_dda` are unbiased
    estimators of it, `raymarch` is a 64-step quadrature (its bias is bounded below); their converged mean over N
    accumulation frames must agree within 3 sigma / sqrt(N).

The scene: a 32^3 stack of constant value (one brighter voxel in a far corner fixes the normalisation at 1/2), clip box
strictly inside the data so that every trilinear tap sees the constant, a transfer function with one colour and one
alpha.  Camera rays come from the closed form of SURVEY Appendix C (tan(fov/2), look-at basis), not from
setup_world_ray; the box from the transform chain of viewer.ts:1089-1099 written out by hand."""
import math

import numpy as np
import pytest

W, H = 24, 16
ALPHA, COLOUR = 0.1, np.array([0.8, 0.5, 0.3])
CLIP_LO, CLIP_HI = 3.0 / 64.0, 29.0 / 64.0     # index 3 .. 29 of the padded 64^3 grid: inside the 32^3 data, taps included
EYE, LOOK = np.array([0.1, 0.0, -0.65]), np.array([-0.27, -0.27, -0.27])   # looks at the middle of the clipped region
LIGHT = np.array([-1.0, -1.0, -1.0]) / math.sqrt(3.0)


def _scene(oracle, mode, **kw):
    from tests.common import make_scene
    vox = np.full((32, 32, 32), 2000, dtype=np.uint16)
    vox[31, 31, 31] = 4000                       # max of the stack: the rest normalises to exactly 1/2
    g = oracle.BrickGrid(vox, (1.0, 1.0, 1.0))
    tf = np.tile(np.array([*COLOUR, ALPHA], dtype=np.float32), (16, 1)).reshape(-1)
    s, cam, vol, ds, p = make_scene(g, W, H, mode, cam_pos=tuple(EYE), look_at=tuple(LOOK),
                                    clip_min=(CLIP_LO,) * 3, clip_max=(CLIP_HI,) * 3, show_environment=False,
                                    use_env=False, light_dir=tuple(LIGHT), **kw)
    return g, tf, 16, p


def _rays(sub=1):
    """world-space camera rays through the pixel grid (closed form); sub x sub positions per pixel covering the
    support of the reference's jitter (+-1 pixel, triangular weights, fragment.frag:146)"""
    aspect, th = W / H, math.tan(math.pi / 6.0)                      # fovy = pi / 3 (scene.ts:55)
    z = (EYE - LOOK) / np.linalg.norm(EYE - LOOK)
    x = np.cross([0.0, 1.0, 0.0], z); x /= np.linalg.norm(x)
    y = np.cross(z, x)
    offs = np.array([0.0]) if sub == 1 else (np.arange(sub) + 0.5) / sub * 2.0 - 1.0     # in pixels
    wts = np.array([1.0]) if sub == 1 else (1.0 - np.abs(offs))
    wts = wts / wts.sum()
    py, px = np.meshgrid(np.arange(H), np.arange(W), indexing="ij")
    dirs, weights = [], []
    for oy, wy in zip(offs, wts):
        for ox, wx in zip(offs, wts):
            nx = ((px + 0.5 + ox) / W) * 2.0 - 1.0
            ny = ((py + 0.5 + oy) / H) * 2.0 - 1.0
            d = x[None, None, :] * (nx * aspect * th)[..., None] + y[None, None, :] * (ny * th)[..., None] - z[None, None, :]
            dirs.append(d / np.linalg.norm(d, axis=-1, keepdims=True))
            weights.append(wx * wy)
    return dirs, weights


def _box():
    """the clipped box in world space: index -> world is (i - 32) / 64 per axis (padded extent 64, spacing 1)"""
    lo = (np.array([CLIP_LO] * 3) * 64.0 - 32.0) / 64.0
    hi = (np.array([CLIP_HI] * 3) * 64.0 - 32.0) / 64.0
    return lo, hi


def _slab(o, d, lo, hi):
    with np.errstate(divide="ignore", invalid="ignore"):
        t0, t1 = (lo - o) / d, (hi - o) / d
    near = np.maximum(0.0, np.minimum(t0, t1).max(axis=-1))
    far = np.maximum(t0, t1).min(axis=-1)
    return near, far


SIGMA = 64.0 * ALPHA            # extinction per world unit: u_volume_maj (= density scale 64) * alpha (viewer.ts:1322)
F_P, LE = 1.0 / (4.0 * math.pi), 4.01


def expected_dvr():
    lo, hi = _box()
    (d,), _ = _rays()
    near, far = _slab(EYE, d, lo, hi)
    length = np.maximum(far - near, 0.0)
    k = 0.9 * F_P * LE                       # albedo * mis (= 1 without the environment term) * f_p * Le / pdf
    return (1.0 - np.exp(-SIGMA * length))[..., None] * COLOUR * k, length


def expected_single_scatter(n_quad=3000):
    lo, hi = _box()
    dirs, wts = _rays(sub=5)
    out = np.zeros((H, W))
    interior = np.ones((H, W), dtype=bool)      # every sub-ray of the pixel's jitter footprint crosses the box
    u = (np.arange(n_quad) + 0.5) / n_quad
    for d, w in zip(dirs, wts):
        near, far = _slab(EYE, d, lo, hi)
        ok = far > near
        interior &= (far - near) > 0.02
        span = np.where(ok, far - near, 0.0)
        t = near[..., None] + span[..., None] * u                               # (H, W, Q)
        pos = EYE + t[..., None] * d[..., None, :]
        _, s_exit = _slab(pos, np.broadcast_to(-LIGHT, pos.shape), lo, hi)      # towards the light (w_i = -light_dir)
        f = SIGMA * np.exp(-SIGMA * (t - near[..., None])) * np.exp(-SIGMA * np.clip(s_exit, 0.0, 10.0))
        out += w * f.mean(axis=-1) * span
    return out[..., None] * (0.9 * COLOUR) * F_P * LE, interior


def test_oracle_dvr_is_beer_lambert(oracle):
    g, tf, L, p = _scene(oracle, "dvr", dvr_step_voxels=0.125)
    img, c = oracle.render(p, g, tf, L)
    want, length = expected_dvr()
    assert (length > 0.2).sum() > 100                                   # the box fills a good part of the image
    dt = 0.125 / 64.0
    # the march covers n dt with |n dt - l| <= dt / 2: relative error of the exponent below sigma dt / 2
    tol = (0.9 * F_P * LE) * COLOUR.max() * (SIGMA * dt * 0.5 + 2e-4) + 1e-6
    assert np.abs(img[..., :3] - want).max() <= tol, float(np.abs(img[..., :3] - want).max())
    assert c.tf_samples == c.samples > 0                                # every sample lies inside the sample range


def _mc_check(mean, sigma, n, want, interior, bias_rel):
    """per pixel on the pixels whose whole jitter footprint crosses the box (on the silhouette the 5 x 5 footprint
    quadrature of the closed form is too coarse for a per-pixel statement); the image mean over ALL pixels"""
    assert interior.sum() >= 100
    bound = 3.0 * sigma / math.sqrt(n) + bias_rel * np.abs(want) + 2e-5
    err = np.abs(mean - want)
    inside = (err <= bound)[interior]
    assert inside.mean() >= 0.99, (float((err / bound)[interior].max()), int((~inside).sum()))   # 3 sigma: ~0.3 % outside
    # the image mean is a much tighter estimate than any single pixel
    m_err = abs(float(mean.mean() - want.mean()))
    m_bound = 4.0 * math.sqrt(float((sigma ** 2).sum()) / n) / sigma.size + bias_rel * float(np.abs(want).mean()) + 1e-5
    assert m_err <= m_bound, (m_err, m_bound)


# raymarch places a collision on its 64-step grid (raymarch.glsl:28-52): the start jitter makes the grid uniform along the
# ray but the sample of an interval stands for its whole optical depth -- a first-order quadrature, bias below 3 %
BIAS = {"default": 0.0, "no_dda": 0.0, "raymarch": 0.03}


@pytest.mark.parametrize("mode", ["default", "no_dda", "raymarch"])
def test_oracle_single_scatter_expectation(oracle, mode):
    g, tf, L, p = _scene(oracle, mode, bounces=1)
    n = 384
    frames = np.stack([oracle.render(p, g, tf, L, frame_index=f)[0][..., :3] for f in range(n)]).astype(np.float64)
    want, interior = expected_single_scatter()
    assert want.max() > 0.01
    _mc_check(frames.mean(axis=0), frames.std(axis=0), n, want, interior, BIAS[mode])
    # in aggregate the estimators reproduce the integral to a fraction of a per cent (measured: 0.999 / 0.996 / 1.0003)
    ratio = frames.mean(axis=0)[interior].sum() / want[interior].sum()
    assert abs(ratio - 1.0) <= 0.01 + BIAS[mode], ratio


@pytest.mark.gpu
def test_hip_dvr_is_beer_lambert(oracle):
    from volxel_amd import Volxel3DRenderer
    import ctypes as C
    g, tf, L, p = _scene(oracle, "dvr", dvr_step_voxels=0.125)
    want, _ = expected_dvr()
    dt = 0.125 / 64.0
    tol = (0.9 * F_P * LE) * COLOUR.max() * (SIGMA * dt * 0.5 + 2e-4) + 1e-6
    for layout in (0, 1, 2):
        r = Volxel3DRenderer(W, H, layout=layout)
        r.setup_from_grid(g)
        r.change_transfer_func(tf, L)
        r._check(r._lib.vx_set_params(r._ctx, C.byref(p)))
        r._check(r._lib.vx_render_frame(r._ctx, 0, 0.0))
        img = r.read_accum()
        assert np.abs(img[..., :3] - want).max() <= tol, layout
        r.close()


@pytest.mark.gpu
@pytest.mark.parametrize("mode", ["default", "no_dda", "raymarch"])
def test_hip_single_scatter_expectation(oracle, mode):
    """the device path against the closed form directly (no oracle in between): N frames accumulated by the viewer's
    running mean on the device; the per-pixel spread from 32 single frames of the same launch sequence"""
    from volxel_amd import Volxel3DRenderer
    import ctypes as C
    g, tf, L, p = _scene(oracle, mode, bounces=1)
    want, interior = expected_single_scatter()
    r = Volxel3DRenderer(W, H)
    r.setup_from_grid(g)
    r.change_transfer_func(tf, L)
    r._check(r._lib.vx_set_params(r._ctx, C.byref(p)))
    singles = []
    for f in range(32):
        r._check(r._lib.vx_render_frame(r._ctx, 1000 + f, 0.0))
        singles.append(r.read_accum()[..., :3].astype(np.float64))
    sigma = np.stack(singles).std(axis=0)
    n = 1024
    w = (C.c_float * n)(*[k / (k + 1.0) for k in range(n)])          # running mean of frames 0 .. n-1
    r._check(r._lib.vx_render_frames(r._ctx, 0, n, w, 32))
    mean = r.read_accum()[..., :3].astype(np.float64)
    _mc_check(mean, sigma * 1.25 + 1e-4, n, want, interior, BIAS[mode])   # (the spread estimate from 32 frames is itself noisy)
    ratio = mean[interior].sum() / want[interior].sum()
    assert abs(ratio - 1.0) <= 0.01 + BIAS[mode], ratio
    r.close()


def test_trilinear_reproduces_a_linear_field(oracle):
    """A5 pins its own conventions on a linear ramp: trilinear interpolation is exact on linear functions, so with voxel
    centres at integer + 1/2 (common.glsl:61-69: fract / floor of p - 1/2) the look-up at a continuous index position p
    must return ramp(p - 1/2) up to the brick quantisation (half a code of the brick's range, range ends in binary16).
    A half-voxel shift, swapped axes or a wrong tap order would be off by tens of codes."""
    L = oracle.lib()
    n = 40
    z, y, x = np.meshgrid(np.arange(n), np.arange(n), np.arange(n), indexing="ij")
    ramp = lambda X, Y, Z: 200.0 + 30.0 * X + 20.0 * Y + 10.0 * Z
    vox = ramp(x, y, z).astype(np.uint16)
    vmax = float(vox.max())
    g = oracle.BrickGrid(vox, (1.0, 1.0, 1.0))
    vol = oracle.make_volume(g)
    rng = np.random.default_rng(3)
    worst = 0.0
    for p in rng.uniform(9.0, 31.0, size=(400, 3)):      # bricks 1..3: their dilated windows see data only (quirk Q4)
        got = L.vxo_lookup_density_trilinear(vol, 1.0, float(p[0]), float(p[1]), float(p[2]))
        want = ramp(p[0] - 0.5, p[1] - 0.5, p[2] - 0.5) / vmax
        worst = max(worst, abs(got - want))
    # a brick's range here spans <= (12 * 30 + 12 * 20 + 12 * 10) / vmax (the dilated 12^3 window, brick.rs:101-103):
    # half a u8 code of it + binary16 rounding of the range ends (2^-11 relative)
    brick_range = 12.0 * 60.0 / vmax
    assert worst <= 0.5 * brick_range / 255.0 + 2.0 ** -11 + 1e-6, worst
    assert worst < 0.25 * 10.0 / vmax * 2.5          # far below what a half-voxel shift along the slowest axis would give (5 / vmax)


def test_trilinear_against_scipy_on_random_data(oracle):
    """A4 + A5 against third-party code on data with no structure: the decoded voxels (vxo_lookup_density_brick, tap by
    tap) interpolated by scipy.ndimage.map_coordinates (order 1, grid-constant zero padding -- the oracle rule for taps beyond the
    volume) must agree with vxo_lookup_density_trilinear at random continuous positions, faces and corners included.
    scipy's interpolation shares no code with the oracle or the kernels: a wrong tap order, weight or offset shows as
    an error of the order of the data's contrast, not of fp32 rounding."""
    ndimage = pytest.importorskip("scipy.ndimage")
    L = oracle.lib()
    rng = np.random.default_rng(11)
    vox = rng.integers(0, 4096, size=(19, 27, 33), dtype=np.uint16)      # ragged: padded to 24 x 32 x 40
    g = oracle.BrickGrid(vox, (1.0, 1.0, 1.0))
    vol = oracle.make_volume(g)
    nz, ny, nx = vox.shape
    ez, ey, ex = [8 * ((n + 7) // 8) for n in (nz, ny, nx)]              # the padded extent the index space spans
    dec = np.zeros((ez, ey, ex), dtype=np.float64)
    for z in range(ez):
        for y in range(ey):
            for x in range(ex):
                dec[z, y, x] = L.vxo_lookup_density_brick(vol, x, y, z)
    pts = rng.uniform(-0.75, 1.0, size=(600, 3)) * np.array([ex, ey, ez]) + rng.uniform(0, 1, size=(600, 3))
    pts = np.concatenate([pts, [[0.0, 0.0, 0.0], [0.5, 0.5, 0.5], [ex, ey, ez], [ex - 0.5, 0.5, ez - 0.5], [nx - 0.25, ny + 0.25, 3.5]]])
    # voxel centres at integer + 1/2: array coordinate = p - 1/2, in (z, y, x) order
    want = ndimage.map_coordinates(dec, [pts[:, 2] - 0.5, pts[:, 1] - 0.5, pts[:, 0] - 0.5], order=1, mode="grid-constant", cval=0.0)
    got = np.array([L.vxo_lookup_density_trilinear(vol, 1.0, float(p[0]), float(p[1]), float(p[2])) for p in pts])
    assert np.abs(dec).max() > 0.5 and np.abs(got).max() > 0.3          # there is contrast to get wrong
    assert np.abs(got - want).max() <= 2e-6, np.abs(got - want).max()
