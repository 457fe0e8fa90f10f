"""Independent NumPy restatement of RNG, brick codec, the DVR loop and -- pixel by pixel -- the three
reference render modes with trace_path (path_pixel).  TEST INFRASTRUCTURE.

Second implementation written against the reference text (not against vx_oracle.c) so that
the two can pin each other -- the reference itself ships no vectors (PARITY UNPINNED).
Vectorised over pixels / bricks; fused multiply-adds are emulated in float64 (exact product,
one extra rounding in the sum: differs from a true fma only on ~2^-29 of operations).
"""
from __future__ import annotations

import numpy as np

U32 = np.uint32
F32 = np.float32


# ---- shaders/random.glsl -------------------------------------------------------------
def tea(v0, v1, n=32):  # random.glsl:41-51
    v0 = np.asarray(v0, dtype=U32).copy()
    v1 = np.asarray(v1, dtype=U32).copy()
    s0 = U32(0)
    with np.errstate(over="ignore"):
        for _ in range(n):
            s0 = U32(s0 + U32(0x9E3779B9))
            v0 += ((v1 << U32(4)) + U32(0xA341316C)) ^ (v1 + s0) ^ ((v1 >> U32(5)) + U32(0xC8013EA4))
            v1 += ((v0 << U32(4)) + U32(0xAD90777D)) ^ (v0 + s0) ^ ((v0 >> U32(5)) + U32(0x7E95761E))
    return v0


def wang(x):  # random.glsl:59-66
    x = np.asarray(x, dtype=U32)
    with np.errstate(over="ignore"):
        x = (x ^ U32(61)) ^ (x >> U32(16))
        x = x * U32(9)
        x = x ^ (x >> U32(4))
        x = x * U32(0x27D4EB2D)
        x = x ^ (x >> U32(15))
    return x


def rotl(x, k):
    return (x << U32(k)) | (x >> U32(32 - k))


class Xoshiro:
    """vectorised state of random.glsl:69-94 (note s.x + s.z)."""

    def __init__(self, seed):
        seed = np.asarray(seed, dtype=U32)
        with np.errstate(over="ignore"):
            self.s = [wang(seed + U32(i)) for i in range(4)]

    def next(self):
        s = self.s
        with np.errstate(over="ignore"):
            result = rotl(s[0] + s[2], 7) + s[0]
        t = s[1] << U32(9)
        s[2] = s[2] ^ s[0]
        s[3] = s[3] ^ s[1]
        s[1] = s[1] ^ s[2]
        s[0] = s[0] ^ s[3]
        s[2] = s[2] ^ t
        s[3] = rotl(s[3], 11)
        return result

    def rng(self):  # random.glsl:103-106
        return (self.next() >> U32(8)).astype(F32) / F32(16777216.0)


def pixel_seed(px, py, res_x, frame):  # fragment.frag:143
    with np.errstate(over="ignore"):
        idx = (np.asarray(py, dtype=U32) * U32(res_x) + np.asarray(px, dtype=U32)) * U32(42)
    return tea(idx, np.full_like(idx, frame, dtype=U32))


# ---- brick codec (dicom_preprocessor/src/brick.rs) --------------------------------------
def brick_count(dims):  # brick.rs:77
    return tuple(int(np.ceil(np.ceil(d / 8) / 8) * 8) for d in dims)


def build_bricks(vox: np.ndarray, max_value=None):
    """Returns dict with indirection, range_packed, atlas, atlas_size, mips, brick_counter.
    Vectorised: padded density volume, 12^3 window min/max by strided views."""
    vz, vy, vx = vox.shape
    mv = int(max_value) if max_value else int(vox.max())
    bc = brick_count((vx, vy, vz))
    ex = tuple(b * 8 for b in bc)
    dens = np.zeros((ex[2] + 4, ex[1] + 4, ex[0] + 4), dtype=F32)  # +-2 apron, zeros outside
    dens[2:2 + vz, 2:2 + vy, 2:2 + vx] = vox.astype(F32) / F32(mv)
    nb = bc[0] * bc[1] * bc[2]
    rmin = np.empty((bc[2], bc[1], bc[0]), dtype=F32)
    rmax = np.empty_like(rmin)
    for bz in range(bc[2]):
        slab = dens[bz * 8:bz * 8 + 12]
        for by in range(bc[1]):
            row = slab[:, by * 8:by * 8 + 12, :]
            # windows along x: 12 wide, stride 8
            w = np.lib.stride_tricks.sliding_window_view(row, 12, axis=2)[:, :, ::8, :]
            rmin[bz, by] = w.min(axis=(0, 1, 3))
            rmax[bz, by] = w.max(axis=(0, 1, 3))
    mn16 = rmin.astype(np.float16)
    mx16 = rmax.astype(np.float16)
    packed = (mn16.view(np.uint16).astype(U32) << U32(16)) | mx16.view(np.uint16).astype(U32)
    nonconst = (rmin != rmax).reshape(-1)
    order = np.cumsum(nonconst) - 1
    counter = int(nonconst.sum())
    slot = np.where(nonconst, order, 0).astype(np.int64)
    px, py, pz = slot % bc[0], (slot // bc[0]) % bc[1], slot // (bc[0] * bc[1])
    ind = np.where(nonconst, (px | (py << 10) | (pz << 20)), 0).astype(U32)
    slices = int(8 * np.round(np.ceil(F32(counter) / F32(bc[0] * bc[1]))))
    atlas = np.zeros((slices, bc[1] * 8, bc[0] * 8), dtype=np.uint8)
    rx = mn16.astype(F32).reshape(-1)
    ry = mx16.astype(F32).reshape(-1)
    core = dens[2:2 + ex[2], 2:2 + ex[1], 2:2 + ex[0]]
    for bi in np.nonzero(nonconst)[0]:
        bz, by, bx = bi // (bc[0] * bc[1]), (bi // bc[0]) % bc[1], bi % bc[0]
        v = core[bz * 8:bz * 8 + 8, by * 8:by * 8 + 8, bx * 8:bx * 8 + 8]
        with np.errstate(divide="ignore", invalid="ignore"):
            n = (v - rx[bi]) / (ry[bi] - rx[bi])
        n = np.where(n < 0, F32(0), n)
        n = np.where(n > 1, F32(1), n)
        r = F32(255) * n
        r = np.where(np.isnan(r), F32(0), np.floor(np.abs(r) + F32(0.5)) * np.sign(r))  # round half away
        q = np.clip(r, 0, 255).astype(np.uint8)
        atlas[pz[bi] * 8:pz[bi] * 8 + 8, py[bi] * 8:py[bi] * 8 + 8, px[bi] * 8:px[bi] * 8 + 8] = q
    mips = []
    smin, smax = mn16.astype(F32), mx16.astype(F32)
    for _ in range(3):
        z, y, x = smin.shape
        a = smin.reshape(z // 2, 2, y // 2, 2, x // 2, 2).min(axis=(1, 3, 5))
        b = smax.reshape(z // 2, 2, y // 2, 2, x // 2, 2).max(axis=(1, 3, 5))
        a16, b16 = a.astype(np.float16), b.astype(np.float16)
        mips.append(((a16.view(np.uint16).astype(U32) << U32(16)) | b16.view(np.uint16).astype(U32)).reshape(-1))
        smin, smax = a16.astype(F32), b16.astype(F32)
    return dict(brick_count=bc, indirection=ind, range_packed=packed.reshape(-1), atlas=atlas.reshape(-1),
                atlas_size=(bc[0] * 8, bc[1] * 8, slices), mips=mips, brick_counter=counter, nb=nb)


# ---- shader-side lookups (sampling/common.glsl) ---------------------------------------
def fma(a, b, c):
    return (np.asarray(a, dtype=np.float64) * np.asarray(b, dtype=np.float64)
            + np.asarray(c, dtype=np.float64)).astype(F32)


class NpVolume:
    def __init__(self, grid):
        self.bc = tuple(grid.indirection_size)
        self.ext = tuple(grid.index_extent)
        self.ind = np.asarray(grid.indirection, dtype=U32)
        r = np.asarray(grid.range, dtype=np.uint16).reshape(-1, 2)
        self.mx = r[:, 0].view(np.float16).astype(F32)
        self.mn = r[:, 1].view(np.float16).astype(F32)
        self.atlas = np.asarray(grid.atlas, dtype=np.uint8)
        self.asz = tuple(grid.atlas_size)

    def brick(self, x, y, z):  # common.glsl:35-43 with OOB -> 0
        x, y, z = [np.asarray(a, dtype=np.int64) for a in (x, y, z)]
        ok = (x >= 0) & (y >= 0) & (z >= 0) & (x < self.ext[0]) & (y < self.ext[1]) & (z < self.ext[2])
        xc, yc, zc = np.where(ok, x, 0), np.where(ok, y, 0), np.where(ok, z, 0)
        bi = ((zc >> 3) * self.bc[1] + (yc >> 3)) * self.bc[0] + (xc >> 3)
        p = self.ind[bi].astype(np.int64)
        ax = ((p & 1023) << 3) + (xc & 7)
        ay = (((p >> 10) & 1023) << 3) + (yc & 7)
        az = (((p >> 20) & 1023) << 3) + (zc & 7)
        inb = az < self.asz[2]
        ai = np.where(inb, (az * self.asz[1] + ay) * self.asz[0] + ax, 0)
        byte = self.atlas[ai] if self.atlas.size else np.zeros_like(ai, dtype=np.uint8)
        un = np.where(inb, byte.astype(F32) / F32(255.0), F32(0))
        mn, mx = self.mn[bi], self.mx[bi]
        return np.where(ok, fma(un, mx - mn, mn), F32(0))

    def trilinear(self, scale, px, py, pz):  # common.glsl:61-69
        return self.trilinear_q(scale, *[np.asarray(a, dtype=F32) - F32(0.5) for a in (px, py, pz)])

    def trilinear_q(self, scale, qx, qy, qz):  # the same from the cell-frame position q = p - 1/2
        q = [np.asarray(a, dtype=F32) for a in (qx, qy, qz)]
        fl = [np.floor(a) for a in q]
        f = [a - b for a, b in zip(q, fl)]
        i = [b.astype(np.int64) for b in fl]

        def mix(a, b, t):
            return fma(b, t, a * (F32(1) - t))

        B = self.brick
        lx0 = mix(B(i[0], i[1], i[2]), B(i[0] + 1, i[1], i[2]), f[0])
        lx1 = mix(B(i[0], i[1] + 1, i[2]), B(i[0] + 1, i[1] + 1, i[2]), f[0])
        hx0 = mix(B(i[0], i[1], i[2] + 1), B(i[0] + 1, i[1], i[2] + 1), f[0])
        hx1 = mix(B(i[0], i[1] + 1, i[2] + 1), B(i[0] + 1, i[1] + 1, i[2] + 1), f[0])
        return F32(scale) * mix(mix(lx0, lx1, f[1]), mix(hx0, hx1, f[1]), f[2])


def transfer(tf, L, sr, d):  # common.glsl:78-83
    tf = np.asarray(tf, dtype=F32).reshape(-1, 4)
    i = np.clip(np.floor(d * F32(L)).astype(np.int64), 0, L - 1)
    out = tf[i]
    bad = (d < F32(sr[0])) | (d > F32(sr[1]))
    return np.where(bad[..., None], F32(0), out)


# ---- DVR over an image (fragment.frag main + SURVEY A12), no jitter -------------------------
def _mat_mul(m, x, y, z, w):
    m = np.asarray(m, dtype=F32)
    return [fma(m[12 + i], w, fma(m[8 + i], z, fma(m[4 + i], y, m[i] * x))) for i in range(4)]


def dvr_image(p, grid, tf, L, max_iter=100000):
    W, H = p.res[0], p.res[1]
    px, py = np.meshgrid(np.arange(W), np.arange(H))
    tex_x = (px.astype(F32) + F32(0.5)) / F32(W)
    tex_y = (py.astype(F32) + F32(0.5)) / F32(H)
    one, zero = np.ones_like(tex_x), np.zeros_like(tex_x)
    cw = _mat_mul(p.camera_view_inv[:], zero, zero, zero, one)
    cam = [cw[i] / cw[3] for i in range(3)]
    vp = _mat_mul(p.camera_proj_inv[:], fma(tex_x, F32(2), F32(-1)), fma(tex_y, F32(2), F32(-1)), zero, one)
    vv = [vp[i] / vp[3] for i in range(3)]
    wp = _mat_mul(p.camera_view_inv[:], vv[0], vv[1], vv[2], one)
    d = [wp[i] / wp[3] - cam[i] for i in range(3)]
    if getattr(p, "camera_ortho", 0):
        # [build] orthographic camera (BASELINE config 1): near-plane point of the pixel, camera -z axis
        npt = _mat_mul(p.camera_proj_inv[:], fma(tex_x, F32(2), F32(-1)), fma(tex_y, F32(2), F32(-1)), -one, one)
        wo = _mat_mul(p.camera_view_inv[:], npt[0] / npt[3], npt[1] / npt[3], npt[2] / npt[3], one)
        cam = [wo[i] / wo[3] for i in range(3)]
        wd = _mat_mul(p.camera_view_inv[:], zero, zero, -one, zero)
        d = [wd[i] for i in range(3)]
    dd = fma(d[2], d[2], fma(d[1], d[1], d[0] * d[0]))
    inv = F32(1) / np.sqrt(dd)
    d = [a * inv for a in d]
    with np.errstate(divide="ignore", invalid="ignore"):
        lo = [(F32(p.volume_aabb_min[i]) - cam[i]) * (F32(1) / d[i]) for i in range(3)]
        hi = [(F32(p.volume_aabb_max[i]) - cam[i]) * (F32(1) / d[i]) for i in range(3)]
    gmin = lambda a, b: np.where(b < a, b, a)
    gmax = lambda a, b: np.where(a < b, b, a)
    tmin = [gmin(a, b) for a, b in zip(lo, hi)]
    tmax = [gmax(a, b) for a, b in zip(lo, hi)]
    near = gmax(zero, gmax(tmin[0], gmax(tmin[1], tmin[2])))
    far = gmin(tmax[0], gmin(tmax[1], tmax[2]))
    hit = near <= far
    ip = _mat_mul(p.density_transform_inv[:], cam[0], cam[1], cam[2], one)
    idr = _mat_mul(p.density_transform_inv[:], d[0], d[1], d[2], zero)
    il = fma(idr[2], idr[2], fma(idr[1], idr[1], idr[0] * idr[0]))
    dt = F32(p.dvr_step_voxels) / np.sqrt(il)
    t0 = fma(F32(0.5), dt, near)
    vol = NpVolume(grid)
    C = [np.zeros_like(tex_x) for _ in range(3)]
    T = np.ones_like(tex_x)
    tau = np.zeros_like(tex_x)
    # [build] march contract: the ray has n = min(ceil((far - t0) / dt), max_steps) samples
    # (0 unless the quotient is positive); sample k sits at q = fma(k, dq, q0) in the cell frame (position - 1/2),
    # dq = dt * idir, q0 = fma(t0, idir, ipos) - 1/2, per axis
    with np.errstate(divide="ignore", invalid="ignore"):
        x = (far - t0) / dt
    n = np.where(x > 0, np.minimum(np.ceil(x), F32(p.dvr_max_steps)), F32(0)).astype(F32)
    n = np.where(hit, n, F32(0))
    dq = [dt * idr[i] for i in range(3)]
    q0 = [fma(t0, idr[i], ip[i]) - F32(0.5) for i in range(3)]
    done_all = np.zeros_like(hit)
    samples = 0
    k = 0
    while k < max_iter:
        alive = (F32(k) < n) & ~done_all
        if not alive.any():
            break
        samples += int(alive.sum())
        q = [fma(F32(k), dq[i], q0[i]) for i in range(3)]
        dens = vol.trilinear_q(p.volume_density_scale, *q)
        rgba = transfer(tf, L, p.sample_range, dens * F32(p.volume_inv_maj))
        a = np.where(alive, rgba[..., 3], F32(0))
        pos_a = a > 0
        tau_n = fma(a * F32(p.volume_maj), dt, tau)
        Tn = np.exp(-tau_n.astype(np.float64)).astype(F32)
        dT = np.where(pos_a, T - Tn, F32(0))
        for c in range(3):
            C[c] = np.where(pos_a, fma(dT, rgba[..., c], C[c]), C[c])
        T = np.where(pos_a, Tn, T)
        tau = np.where(pos_a, tau_n, tau)
        done = pos_a & (tau >= F32(p.dvr_ert_tau))
        T = np.where(done, F32(0), T)
        done_all |= done
        k += 1
    nl = [-F32(p.light_dir[i]) for i in range(3)]
    cdot = gmax(fma(d[2], nl[2], fma(d[1], nl[1], d[0] * nl[0])), zero)
    s = np.clip(np.power(cdot.astype(np.float64), 300.0), 0, 1).astype(F32)
    env = F32(p.env_strength) * fma(s, F32(4), F32(0.01))
    out = np.zeros((H, W, 4), dtype=F32)
    for c in range(3):
        Lc = C[c] * F32(p.dvr_gain[c])
        if p.show_environment > 0:
            Lc = np.where(T > 0, fma(T, env, Lc), Lc)
        out[..., c] = Lc
    out[..., 3] = 1
    return out, samples


# ---- the path tracer for one pixel, NO_DDA mode (fragment.frag main + trace_path, ---------------
# ---- sampling/normal.glsl, environment.glsl directional branch), scalar Python -------------------
def _f(x):
    return F32(x)


def _dot(a, b):  # dot as an fma chain
    return fma(a[2], b[2], fma(a[1], b[1], a[0] * b[0]))


def _normalize(a):
    inv = _f(1) / np.sqrt(_dot(a, a))
    return [x * inv for x in a]


def _slab(o, d, lo, hi):  # utils.glsl:61-69
    with np.errstate(divide="ignore", invalid="ignore"):
        inv = [_f(1) / x for x in d]
        a = [(_f(lo[i]) - o[i]) * inv[i] for i in range(3)]
        b = [(_f(hi[i]) - o[i]) * inv[i] for i in range(3)]
    mn = lambda x, y: y if y < x else x
    mx = lambda x, y: y if x < y else x
    tmin = [mn(x, y) for x, y in zip(a, b)]
    tmax = [mx(x, y) for x, y in zip(a, b)]
    near = mx(_f(0), mx(tmin[0], mx(tmin[1], tmin[2])))
    far = mn(tmax[0], mn(tmax[1], tmax[2]))
    return bool(near <= far), near, far


def _logf(x):
    return _f(np.log(np.float64(x)))


def path_pixel(p, grid, tf, L, px, py, frame, env=None):
    """One pixel of fragment.frag:128-158 with u_sample_weight = 0 for the three reference render modes
    (p.render_mode 0 default/DDA, 1 no_dda, 2 raymarch); directional light, or -- with u_use_env = 1 --
    env = (texture [H,W,4] in GL row order, [importance level 0 .. 9]) as environment.ts builds them.
    Returns (rgba, number of volume samples)."""
    assert p.debug_hits == 0 and p.render_mode in (0, 1, 2) and (p.use_env == 0 or env is not None)
    vol = NpVolume(grid)
    # range texture levels 0..3 (.x = R = max), brick.rs:19-23,153-190
    levels = [(np.asarray(grid.range, dtype=np.uint16).reshape(-1, 2)[:, 0].view(np.float16).astype(F32),
               tuple(grid.indirection_size))]
    for m, st in grid.range_mipmaps:
        levels.append((np.asarray(m, dtype=np.uint16).reshape(-1, 2)[:, 0].view(np.float16).astype(F32), tuple(st)))
    W, H = p.res[0], p.res[1]
    rs = Xoshiro(pixel_seed(px, py, W, frame))
    rng = lambda: _f(rs.rng())
    n_samples = 0
    # fragment.frag:146 + :57-65 setup_world_ray(tex, (rng2 + rng2) / 2)
    a0, a1, b0, b1 = rng(), rng(), rng(), rng()
    jx, jy = (a0 + b0) / _f(2), (a1 + b1) / _f(2)
    tex = [(_f(px) + _f(0.5)) / _f(W), (_f(py) + _f(0.5)) / _f(H)]
    off = [fma(jx, _f(2), _f(-1)) * (_f(1) / _f(W)), fma(jy, _f(2), _f(-1)) * (_f(1) / _f(H))]
    sx, sy = tex[0] + off[0], tex[1] + off[1]
    one, zero = _f(1), _f(0)
    cw = _mat_mul(p.camera_view_inv[:], zero, zero, zero, one)
    cam = [cw[i] / cw[3] for i in range(3)]
    vp = _mat_mul(p.camera_proj_inv[:], fma(sx, _f(2), _f(-1)), fma(sy, _f(2), _f(-1)), zero, one)
    vv = [vp[i] / vp[3] for i in range(3)]
    wp = _mat_mul(p.camera_view_inv[:], vv[0], vv[1], vv[2], one)
    o = cam
    d = _normalize([wp[i] / wp[3] - cam[i] for i in range(3)])
    lo, hi = p.volume_aabb_min, p.volume_aabb_max
    inv_maj, maj = _f(p.volume_inv_maj), _f(p.volume_maj)
    albedo = [_f(p.volume_albedo[i]) for i in range(3)]
    g = _f(p.volume_phase_g)
    inv_4pi = _f(1) / (_f(4) * _f(np.pi))

    def to_index(o, d):
        ip = _mat_mul(p.density_transform_inv[:], o[0], o[1], o[2], one)
        idr = _mat_mul(p.density_transform_inv[:], d[0], d[1], d[2], zero)
        return ip[:3], idr[:3]

    def lookup(ip, idr, t):
        nonlocal n_samples
        n_samples += 1
        pos = [fma(t, idr[i], ip[i]) for i in range(3)]
        dens = vol.trilinear(p.volume_density_scale, *pos)
        return transfer(tf, L, p.sample_range, np.asarray(dens * inv_maj))

    def rnd_half_away(x):  # GLSL round(); halves go away from zero (quirk Q12)
        return int(np.floor(np.float64(x) + 0.5)) if x >= 0 else -int(np.floor(-np.float64(x) + 0.5))

    def lookup_tf(dn):
        return transfer(tf, L, p.sample_range, np.asarray(dn))

    def majorant_at(curr, mip):  # common.glsl:50-53, then the extinction of that density (dda.glsl:38)
        sh = 3 + mip
        b = [int(np.floor(c)) >> sh for c in curr]
        data, (sx, sy, sz) = levels[mip]
        r = zero
        if 0 <= b[0] < sx and 0 <= b[1] < sy and 0 <= b[2] < sz:
            r = data[(b[2] * sy + b[1]) * sx + b[0]]
        m = _f(p.volume_density_scale) * r
        return maj * lookup_tf(m * inv_maj)[3]

    def step_dda(pos, inv_dir, mip):  # dda.glsl:11-16
        dim = _f(8 << mip)
        inv_dim = one / dim
        out = []
        for c in range(3):
            off = dim + _f(0.5) if inv_dir[c] >= 0 else _f(-0.5)
            out.append(((np.floor(pos[c] * inv_dim) * dim + off) - pos[c]) * inv_dir[c])
        mn = lambda x, y: y if y < x else x
        return mn(out[0], mn(out[1], out[2]))

    def density_stochastic(pos):  # common.glsl:9-32,72-76
        nonlocal n_samples
        n_samples += 1
        q = [c - _f(0.5) for c in pos]
        ii = [int(np.floor(c)) for c in q]
        t = [q[c] - _f(ii[c]) for c in range(3)]
        t2 = [x * x for x in t]
        sixth = one / _f(6)
        idx = [0, 0, 0]
        w = [sixth * (fma(_f(-3), t[c], fma(_f(3), t2[c], -t[c] * t2[c])) + one) for c in range(3)]
        sw = list(w)
        taps = (lambda c: sixth * (fma(_f(-6), t2[c], _f(3) * t[c] * t2[c]) + _f(4)),
                lambda c: sixth * (fma(_f(3), t[c], fma(_f(3), t2[c], _f(-3) * t[c] * t2[c])) + one),
                lambda c: sixth * t[c] * t2[c])
        for k, wf in enumerate(taps):
            w = [wf(c) for c in range(3)]
            sw = [w[c] + sw[c] for c in range(3)]
            r = [rng(), rng(), rng()]
            for c in range(3):
                den = sw[c] if _f(1e-3) < sw[c] else _f(1e-3)
                if r[c] < w[c] / den:
                    idx[c] = k + 1
        tap = [ii[c] + idx[c] - 1 for c in range(3)]
        return _f(p.volume_density_scale) * vol.brick(tap[0], tap[1], tap[2])

    def sample_raymarch(o, d, thr):  # raymarch.glsl:25-55
        hit, near, far = _slab(o, d, lo, hi)
        if not hit:
            return False, zero
        ip, idr = to_index(o, d)
        tau_target = -_logf(one - rng())
        dt = (far - near) / _f(64)
        near = fma(rng(), dt, near)
        tau, t = zero, zero
        for i in range(64):
            tt = fma(_f(i), dt, near)
            t = far if far < tt else tt
            dens = density_stochastic([fma(t, idr[c], ip[c]) for c in range(3)])
            rgba = lookup_tf(dens * inv_maj)
            tau = fma(rgba[3] * maj, dt, tau)
            if tau >= tau_target:
                for c in range(3):
                    thr[c] = thr[c] * (rgba[c] * albedo[c])
                return True, t
        return False, t

    def transmittance_raymarch(o, d):  # raymarch.glsl:8-23
        hit, near, far = _slab(o, d, lo, hi)
        if not hit:
            return one
        ip, idr = to_index(o, d)
        dt = (far - near) / _f(64)
        near = fma(rng(), dt, near)
        tau = zero
        for i in range(64):
            tt = fma(_f(i), dt, near)
            t = far if far < tt else tt
            dens = density_stochastic([fma(t, idr[c], ip[c]) for c in range(3)])
            tau = fma(lookup_tf(dens * inv_maj)[3] * maj, dt, tau)
        return _f(np.exp(-np.float64(tau)))

    def dda_walk(o, d, shadow, thr):  # dda.glsl:21-62 (shadow) / :65-98 (primary)
        hit, near, far = _slab(o, d, lo, hi)
        if not hit:
            return (one if shadow else False), zero
        ip, idr = to_index(o, d)
        with np.errstate(divide="ignore"):
            ri = [one / x for x in idr]
        t = near + _f(1e-6)
        tr, tau, mip, step = one, -_logf(one - rng()), _f(3), 0
        while t < far and (not shadow or step < 100):
            step += 1
            curr = [fma(t, idr[c], ip[c]) for c in range(3)]
            m = rnd_half_away(mip)
            majorant = majorant_at(curr, m)
            dt = step_dda(curr, ri, m)
            t = t + dt
            tau = fma(-majorant, dt, tau)
            mip = mip + _f(0.25) if mip + _f(0.25) < _f(3) else _f(3)
            if tau > 0:
                continue
            with np.errstate(divide="ignore", invalid="ignore"):
                t = t + tau / majorant
            if t >= far:
                break
            rgba = lookup(ip, idr, t)
            dd = maj * rgba[3]
            if rng() * majorant < dd:
                if not shadow:
                    for c in range(3):
                        thr[c] = (thr[c] * albedo[c]) * rgba[c]
                    return True, t
                with np.errstate(divide="ignore", invalid="ignore"):
                    f = one - maj / majorant
                tr = tr * (f if zero < f else zero)
                if tr < _f(0.1):
                    prob = one - tr
                    if rng() < prob:
                        return zero, t
                    tr = tr / (one - prob)
            tau = -_logf(one - rng())
            mip = mip - _f(2) if zero < mip - _f(2) else zero
        return (tr if shadow else False), t

    def sample_volume(o, d, thr):
        if p.render_mode == 2:
            return sample_raymarch(o, d, thr)
        if p.render_mode == 0:
            return dda_walk(o, d, False, thr)
        return sample_simple(o, d, thr)

    def transmittance(o, d):
        if p.render_mode == 2:
            return transmittance_raymarch(o, d)
        if p.render_mode == 0:
            return dda_walk(o, d, True, None)[0]
        return transmittance_simple(o, d)

    def sample_simple(o, d, thr):  # normal.glsl:33-57
        hit, near, far = _slab(o, d, lo, hi)
        if not hit:
            return False, zero
        ip, idr = to_index(o, d)
        t = fma(-_logf(one - rng()), inv_maj, near)
        while t < far:
            rgba = lookup(ip, idr, t)
            p_real = (maj * rgba[3]) * inv_maj
            if rng() < p_real:
                for c in range(3):
                    thr[c] = thr[c] * (rgba[c] * albedo[c])
                return True, t
            t = fma(-_logf(one - rng()), inv_maj, t)
        return False, t

    def transmittance_simple(o, d):  # normal.glsl:6-31
        hit, near, far = _slab(o, d, lo, hi)
        if not hit:
            return one
        ip, idr = to_index(o, d)
        t = fma(-_logf(one - rng()), inv_maj, near)
        tr = one
        while t < far:
            rgba = lookup(ip, idr, t)
            dd = maj * rgba[3]
            tr = tr * fma(-dd, inv_maj, one)
            if tr < _f(0.1):
                prob = one - tr
                if rng() < prob:
                    return zero
                tr = tr / (one - prob)
            t = fma(-_logf(one - rng()), inv_maj, t)
        return tr

    def phase_hg(cos_t):  # utils.glsl:121-124
        denom = fma(_f(2) * g, cos_t, one + g * g)
        return inv_4pi * (one - g * g) / (denom * np.sqrt(denom))

    def env_texture(u, v):  # texture(u_envmap, uv): LINEAR, REPEAT s, CLAMP_TO_EDGE t, exact fp32 weights
        tex = env[0]
        h, w = tex.shape[:2]
        x, y = fma(u, _f(w), _f(-0.5)), fma(v, _f(h), _f(-0.5))
        fx, fy = np.floor(x), np.floor(y)
        a, b = x - fx, y - fy
        i0, j0 = int(fx), int(fy)
        i1, j1 = (i0 + 1) % w, min(max(j0 + 1, 0), h - 1)
        i0, j0 = i0 % w, min(max(j0, 0), h - 1)
        mix = lambda p0, p1, t: fma(p1, t, p0 * (one - t))
        return [mix(mix(tex[j0, i0, c], tex[j0, i1, c], a), mix(tex[j1, i0, c], tex[j1, i1, c], a), b) for c in range(3)]

    def imp(x, y, mip):
        lv = env[1][mip]
        n = lv.shape[0]
        return lv[y, x] if 0 <= x < n and 0 <= y < n else zero

    def sample_env(u0, u1):  # environment.glsl:35-79
        pxl, pyl, sx_, sy_ = 0, 0, u0, u1
        for mip in range(8, -1, -1):
            pxl, pyl = pxl * 2, pyl * 2
            w = [imp(pxl, pyl, mip), imp(pxl + 1, pyl, mip), imp(pxl, pyl + 1, mip), imp(pxl + 1, pyl + 1, mip)]
            q = [w[0] + w[2], w[1] + w[3]]
            with np.errstate(divide="ignore", invalid="ignore"):
                den = q[0] + q[1]
                dd = q[0] / (den if _f(1e-8) < den else _f(1e-8))
                if sx_ < dd:
                    offx, sx_ = 0, sx_ / dd
                else:
                    offx, sx_ = 1, (sx_ - dd) / (one - dd)
                pxl += offx
                e = w[offx] / q[offx]
                if sy_ < e:
                    sy_ = sy_ / e
                else:
                    pyl, sy_ = pyl + 1, (sy_ - e) / (one - e)
        inv_dim = one / _f(512)
        uvx, uvy = (_f(pxl) + sx_) * inv_dim, (_f(pyl) + sy_) * inv_dim
        clamp01 = lambda x: min(max(x, zero), one)
        theta = clamp01(one - uvy) * _f(np.pi)
        phi = (clamp01(uvx) * _f(2) - one) * _f(np.pi)
        sin_t = _f(np.sin(np.float64(theta)))
        wi = [sin_t * _f(np.cos(np.float64(phi))), _f(np.cos(np.float64(theta))), sin_t * _f(np.sin(np.float64(phi)))]
        t = env_texture(uvx, uvy)
        pdf = imp(pxl, pyl, 0) / imp(0, 0, 9)
        return wi, [_f(p.env_strength) * c for c in t], pdf * inv_4pi

    def lookup_env_map(dr):  # environment.glsl:23-26
        u = _f(np.arctan2(np.float64(dr[2]), np.float64(dr[0]))) / (_f(2) * _f(np.pi)) + _f(0.5)
        v = one - _f(np.arccos(np.float64(dr[1]))) / _f(np.pi)
        return [_f(p.env_strength) * c for c in env_texture(u, v)]

    def lookup_env(dr):  # environment.glsl:19-22 with the pow base clamped at 0 (quirk Q16)
        if p.use_env > 0:
            return lookup_env_map(dr)
        nl = [-_f(p.light_dir[i]) for i in range(3)]
        c = _dot(dr, nl)
        c = c if c > 0 else zero
        s = _f(min(max(float(np.float64(c) ** 300.0), 0.0), 1.0))
        e = _f(p.env_strength) * fma(s, _f(4), _f(0.01))
        return [e, e, e]

    # trace_path, fragment.frag:79-124
    Lr, thr = [zero, zero, zero], [one, one, one]
    free_path, n_paths, f_p = True, 0, zero
    while True:
        hit, t = sample_volume(o, d, thr)
        if not hit:
            break
        o = [fma(t, d[i], o[i]) for i in range(3)]
        e0, e1 = rng(), rng()                                 # rng2 handed to sample_environment
        w_i = [-_f(p.light_dir[i]) for i in range(3)]
        le3, pdf = [_f(p.env_strength) * _f(4.01)] * 3, one
        if p.use_env > 0:
            w_i, le3, pdf = sample_env(e0, e1)
        if pdf > 0:
            f_p = phase_hg(_dot([-x for x in d], w_i))
            mis = (pdf * pdf) / (pdf * pdf + f_p * f_p) if p.show_environment > 0 else one
            tr = transmittance(o, w_i)
            for c in range(3):
                Lr[c] = Lr[c] + thr[c] * mis * f_p * tr * le3[c] / pdf
        n_paths += 1
        if n_paths >= p.bounces:
            free_path = False
            break
        rr = _dot(thr, [_f(0.212671), _f(0.715160), _f(0.072169)])
        if rr < _f(0.1):
            prob = one - rr
            if rng() < prob:
                free_path = False
                break
            thr = [x / (one - prob) for x in thr]
        u0, u1 = rng(), rng()
        if abs(g) < _f(1e-4):
            cos_t = fma(_f(-2), u0, one)
        else:
            q = (one - g * g) / fma(_f(2) * g, u0, one - g)
            cos_t = ((one + g * g) - q * q) / (_f(2) * g)
        sin_t = np.sqrt(max(zero, one - cos_t * cos_t))
        phi = _f(2) * _f(np.pi) * u1
        v = [sin_t * _f(np.cos(np.float64(phi))), sin_t * _f(np.sin(np.float64(phi))), cos_t]
        N = d
        if abs(N[0]) > abs(N[1]):
            ln = np.sqrt(fma(N[2], N[2], N[0] * N[0]))
            T = [-N[2] / ln, zero / ln, N[0] / ln]
        else:
            ln = np.sqrt(fma(N[2], N[2], N[1] * N[1]))
            T = [zero / ln, N[2] / ln, -N[1] / ln]
        B = [fma(N[1], T[2], -(N[2] * T[1])), fma(N[2], T[0], -(N[0] * T[2])), fma(N[0], T[1], -(N[1] * T[0]))]
        sd = _normalize([fma(v[2], N[i], fma(v[1], B[i], v[0] * T[i])) for i in range(3)])
        f_p = phase_hg(_dot([-x for x in d], sd))
        d = sd
    if free_path and p.show_environment > 0:
        le = lookup_env(d)
        pe = zero
        if p.use_env > 0:  # environment.glsl:82-86
            pe = _dot(le, [_f(0.212671), _f(0.715160), _f(0.072169)]) / imp(0, 0, 9) * inv_4pi
        with np.errstate(invalid="ignore", divide="ignore"):
            mis = (f_p * f_p) / (f_p * f_p + pe * pe) if n_paths > 0 else one
        for c in range(3):
            Lr[c] = fma(thr[c] * mis, le[c], Lr[c])
    out = [x if np.isfinite(x) else zero for x in Lr]       # sanitize, utils.glsl:96-98
    return np.array(out + [one], dtype=F32), n_samples
