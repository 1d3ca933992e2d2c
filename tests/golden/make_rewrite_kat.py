#!/usr/bin/env python3
"""Generates tests/golden/rewrite_kat.npz from the REFERENCE ITSELF, run in the container that holds /root/reference:

    make -C oracle ref                       # builds oracle/_ref/rewrite_kat: smallpt_rewrite.cpp #included unmodified behind a harness
    python tests/golden/make_rewrite_kat.py

smallpt2pbrt/smallpt_rewrite.cpp is the pbrt-style fp64 step between smallpt and ky.cpp, and the one reference translation unit this
image builds.  Several of ky.cpp's hot-path formulas are still the ones that file has; the harness (oracle/rewrite_kat.cpp) calls the
reference's own classes on the inputs made here -- every input is a float32 value widened to double, so that ky.cpp's fp32
restatement (oracle/ky_oracle.cpp) and the HIP path can be given the very same numbers -- and this script stores inputs and fp64
outputs.  tests/test_rewrite_kat.py compares them.

Formula by formula (checked against ky.cpp before relying on it):
  kept    Frame / frame_t                  smallpt_rewrite.cpp:122-176 == ky.cpp:526-578 (SetFromZ: same axis choice, same two cross products)
  kept    Sphere::Intersect                706-786 == 1336-1393 except epsilon 1e-4 instead of 1e-3 (1093): inputs keep both roots away
                                           from (0, 2e-3), so the acceptance tests agree
  kept    PerspectiveCamera                651-694 == 1864-1892 except (a) ky normalises front and up in the ctor (the inputs here are unit
                                           already), (b) the rewrite starts its rays 140 units along the direction: only directions compare
  kept    LambertionReflection f / Pdf     873-905 == 2227-2240 where wo and wi share a hemisphere; ky.cpp returns f = 0 across
                                           hemispheres (2232), the rewrite R / pi everywhere: only same-hemisphere rows compare for f
  kept    SpecularReflection::Sample_f     907-934 == 2292-2307
  kept    GammaEncoding                    494 == 1548
  kept    AreaLight::Le                    1114-1117 == 2957-2960 (areal_radiance: the light's radiance where dot(normal, wo) > 0, else black), reached the way a path
                                           ray reaches it: Primitive::Intersect 1135-1146 == surface_t::intersect 3077-3088 (round 5)
  kept    Scene::Intersect                 1184-1197 == 3172-3184: every primitive is tested, ray.distance shrinks, `distance < ray.distance` is strict, so of two
                                           surfaces at exactly the same distance the EARLIER list entry stays (round 5)
  differs CosineSampleHemisphere           the lift z = sqrt(max(0, 1 - x^2 - y^2)) is shared (259-265 == 737-745) but the disk mapping is
                                           polar there (252-257) and concentric in ky.cpp (710-733): the sampled directions differ
  differs FresnelSpecular                  Schlick's approximation there (1003-1008), the exact dielectric Fresnel in ky.cpp (1963-1996)
  differs RecursionPathIntegrater          no next-event estimation, no MIS, no lights besides emissive spheres (1335-1381)
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
EXE = os.path.join(ROOT, "oracle", "_ref", "rewrite_kat")


def unit(v):
    return v / np.linalg.norm(v, axis=-1, keepdims=True)


def f32(a):
    """float32 values (what ky.cpp computes with), returned as float64 so the reference reads exactly those numbers"""
    return np.asarray(a, np.float32).astype(np.float64)


def main():
    if not os.path.exists(EXE):
        sys.exit("oracle/_ref/rewrite_kat is missing: run `make -C oracle ref` where /root/reference exists")
    rng = np.random.default_rng(20261002)

    # Frame: unit normals (some almost along x: the |n.x| > 0.99 branch) and arbitrary vectors
    n_frame = 768
    normals = unit(rng.normal(size=(n_frame, 3)))
    normals[:128] = unit(np.array([1.0, 0.0, 0.0]) * rng.choice([-1, 1], (128, 1)) + 0.2 * rng.normal(size=(128, 3)))
    frame_in = f32(np.concatenate([f32(unit(f32(normals))), rng.uniform(-2, 2, (n_frame, 3))], 1))

    # Sphere::Intersect: the spheres of ky's two scenes' sizes, rays from outside and inside; roots near the epsilons are removed below
    n_sph = 1536
    center = rng.uniform(-1, 1, (n_sph, 3))
    radius = rng.choice([0.5, 0.8, 0.1, 0.3, 0.9], n_sph)
    origin = center + rng.uniform(0.2, 4.0, (n_sph, 1)) * unit(rng.normal(size=(n_sph, 3))) * radius[:, None]
    target = center + rng.uniform(-1.3, 1.3, (n_sph, 3)) * radius[:, None]
    direction = f32(unit(f32(target - origin)))
    direction = f32(direction / np.linalg.norm(direction, axis=1, keepdims=True))
    tmax = np.where(rng.uniform(size=n_sph) < 0.25, rng.uniform(0.05, 3.0, n_sph), np.inf)
    sphere_in = f32(np.concatenate([center, radius[:, None], origin, direction, tmax[:, None]], 1))
    c, r, o, d = sphere_in[:, 0:3], sphere_in[:, 3], sphere_in[:, 4:7], sphere_in[:, 7:10]
    oc = c - o
    nb = (oc * d).sum(1)
    disc = nb * nb - (oc * oc).sum(1) + r * r
    sq = np.sqrt(np.maximum(disc, 0))
    near_eps = lambda t: (t > -1e-3) & (t < 3e-3)                      # noqa: E731  (the two sources' epsilons are 1e-4 and 1e-3)
    keep = ~(near_eps(nb - sq) | near_eps(nb + sq)) & (np.abs(disc) > 1e-4) & (np.abs(nb - sq - sphere_in[:, 10]) > 1e-3) & (np.abs(nb + sq - sphere_in[:, 10]) > 1e-3)
    sphere_in = sphere_in[keep]

    # cameras: the literals of create_cornell_box_scene (ky.cpp:3260-3264) and create_mis_scene (3455-3458); front and up normalised
    # here (ky.cpp's ctor does it, 1869-1870; the rewrite expects a unit front)
    cams = [
        ((-0.0439815, 4.12529, 0.222539), (0.00688625, -0.998505, -0.0542161), (3.73896e-4, -0.0542148, 0.998529), 80.0, (256, 256)),
        ((-0.0439815, 4.12529, 0.222539), (0.00688625, -0.998505, -0.0542161), (3.73896e-4, -0.0542148, 0.998529), 80.0, (1024, 768)),
        ((0.0, 2.0, -15.0), (0.0, -4.0, 12.5), (0.0, 1.0, 0.0), 50.0, (1280, 720)),
    ]
    cam_records, cam_pfilm = [], []
    for pos, front, up, fov, res in cams:
        pf = f32(rng.uniform([0, 0], res, (192, 2)))
        front32 = f32(unit(f32(np.array(front))))
        up32 = f32(unit(f32(np.array(up))))
        cam_records.append(np.concatenate([f32(pos), front32, up32, [fov, res[0], res[1], len(pf)], pf.ravel()]))
        cam_pfilm.append(pf)

    # BSDFs: normal, wo, wi, R
    n_bsdf = 768
    bn = f32(unit(f32(unit(rng.normal(size=(n_bsdf, 3))))))
    wo = unit(rng.normal(size=(n_bsdf, 3)))
    wo[: n_bsdf // 2] = unit(wo[: n_bsdf // 2] + 1.5 * bn[: n_bsdf // 2])
    wi = unit(rng.normal(size=(n_bsdf, 3)))
    R = rng.uniform(0.05, 1.0, (n_bsdf, 3))
    bsdf_in = f32(np.concatenate([bn, f32(unit(f32(wo))), f32(unit(f32(wi))), R], 1))

    lift_in = f32(rng.uniform(size=(256, 2)))
    gamma_in = f32(np.concatenate([rng.uniform(-0.2, 1.2, 1000), np.linspace(0, 1, 513), [0.0, 1.0, 0.5]]))

    # round 5 (drawn AFTER everything above: the earlier sections of the fixture keep their values).
    # AreaLight::Le through Primitive::Intersect: one emitting sphere per row; half of the rays start inside it (dot(normal, wo) < 0: black)
    n_le = 384
    le_c = rng.uniform(-1, 1, (n_le, 3))
    le_r = rng.choice([0.5, 0.1, 0.8], n_le)
    inside = rng.uniform(size=n_le) < 0.5
    le_o = le_c + np.where(inside[:, None], rng.uniform(0.0, 0.8, (n_le, 1)), rng.uniform(1.3, 4.0, (n_le, 1))) * unit(rng.normal(size=(n_le, 3))) * le_r[:, None]
    le_target = le_c + rng.uniform(-1.2, 1.2, (n_le, 3)) * le_r[:, None]
    le_d = f32(unit(f32(le_target - le_o)))
    le_d = f32(le_d / np.linalg.norm(le_d, axis=1, keepdims=True))
    le_L = rng.uniform(0.5, 30.0, (n_le, 3))
    le_in = f32(np.concatenate([le_c, le_r[:, None], le_L, le_o, le_d], 1))
    c, r, o, d = le_in[:, 0:3], le_in[:, 3], le_in[:, 7:10], le_in[:, 10:13]
    oc = c - o
    nb = (oc * d).sum(1)
    disc = nb * nb - (oc * oc).sum(1) + r * r
    sq = np.sqrt(np.maximum(disc, 0))
    le_in = le_in[~(near_eps(nb - sq) | near_eps(nb + sq)) & (np.abs(disc) > 1e-4)]
    # Scene::Intersect: three lists of spheres; in each, two spheres appear TWICE (bit-identical records at different list positions: exact ties, the
    # earlier entry must win), and rays start outside every sphere
    scene_sets, scene_rays = [], []
    for n_s in (6, 9, 12):
        cs = rng.uniform(-1.2, 1.2, (n_s - 2, 3))
        rs = rng.uniform(0.15, 0.5, n_s - 2)
        spheres = f32(np.concatenate([cs, rs[:, None]], 1))
        order = list(range(n_s - 2))
        order.insert(int(rng.integers(1, n_s - 2)), 0)          # sphere 0 again, later in the list
        order.append(int(rng.integers(1, n_s - 2)))             # another one again, last
        spheres = spheres[order]
        n_r = 384
        ro = 3.0 * unit(rng.normal(size=(n_r, 3))) * rng.uniform(1.0, 1.5, (n_r, 1))
        rt = spheres[rng.integers(0, n_s, n_r), 0:3] + rng.uniform(-0.6, 0.6, (n_r, 3))
        rd = f32(unit(f32(rt - ro)))
        rd = f32(rd / np.linalg.norm(rd, axis=1, keepdims=True))
        tm = np.where(rng.uniform(size=n_r) < 0.2, rng.uniform(1.0, 4.0, n_r), np.inf)
        rays = f32(np.concatenate([ro, rd, tm[:, None]], 1))
        # no root of any sphere near the two sources' epsilons or near tmax, no grazing hit (a flag that fp32 and fp64 could decide differently)
        ok = np.ones(n_r, bool)
        for sph in spheres:
            oc = sph[0:3] - rays[:, 0:3]
            nb = (oc * rays[:, 3:6]).sum(1)
            disc = nb * nb - (oc * oc).sum(1) + sph[3] ** 2
            sq = np.sqrt(np.maximum(disc, 0))
            ok &= (np.abs(disc) > 1e-3) & ~near_eps(nb - sq) & ~near_eps(nb + sq) & (np.abs(nb - sq - rays[:, 6]) > 1e-3) & (np.abs(nb + sq - rays[:, 6]) > 1e-3)
        rays = rays[ok]
        scene_sets.append(spheres)
        scene_rays.append(rays)
    scene_records = [np.concatenate([[len(sp)], sp.ravel(), [len(ry)], ry.ravel()]) for sp, ry in zip(scene_sets, scene_rays)]

    counts = np.array([len(frame_in), len(sphere_in), len(cams), len(bsdf_in), len(lift_in), len(gamma_in), len(le_in), len(scene_sets)], np.int64)
    payload = np.concatenate([frame_in.ravel(), sphere_in.ravel()] + cam_records + [bsdf_in.ravel(), lift_in.ravel(), gamma_in.ravel(), le_in.ravel()] + scene_records).astype("<f8")
    with tempfile.TemporaryDirectory() as tmp:
        fin, fout = os.path.join(tmp, "in.bin"), os.path.join(tmp, "out.bin")
        with open(fin, "wb") as fh:
            fh.write(counts.astype("<i8").tobytes())
            fh.write(payload.tobytes())
        subprocess.check_call([EXE, fin, fout])
        out = np.fromfile(fout, "<f8")
    pos = 0

    def take(n, w):
        nonlocal pos
        a = out[pos:pos + n * w].reshape(n, w)
        pos += n * w
        return a

    frame_out = take(len(frame_in), 15)
    sphere_out = take(len(sphere_in), 8)
    cam_out = [take(len(pf), 3) for pf in cam_pfilm]
    bsdf_out = take(len(bsdf_in), 11)
    lift_out = take(len(lift_in), 3)
    gamma_out = take(len(gamma_in), 1)[:, 0]
    le_out = take(len(le_in), 5)
    scene_out = [take(len(ry), 6) for ry in scene_rays]
    assert pos == out.size
    path = os.path.join(ROOT, "tests", "golden", "rewrite_kat.npz")
    np.savez_compressed(
        path, frame_in=frame_in.astype(np.float32), frame_out=frame_out, sphere_in=sphere_in.astype(np.float32), sphere_out=sphere_out,
        cam_pfilm=np.stack(cam_pfilm).astype(np.float32), cam_out=np.stack(cam_out), cam_res=np.array([c[4] for c in cams], np.int32),
        bsdf_in=bsdf_in.astype(np.float32), bsdf_out=bsdf_out, lift_in=lift_in.astype(np.float32), lift_out=lift_out,
        gamma_in=gamma_in.astype(np.float32), gamma_out=gamma_out.astype(np.uint8),
        le_in=le_in.astype(np.float32), le_out=le_out,
        **{"scene%d_spheres" % i: sp.astype(np.float32) for i, sp in enumerate(scene_sets)},
        **{"scene%d_rays" % i: ry.astype(np.float32) for i, ry in enumerate(scene_rays)},
        **{"scene%d_out" % i: so for i, so in enumerate(scene_out)})
    ties = sum(int(((so[:, 0] == 1) & np.isin(so[:, 2], [0])).sum()) for so in scene_out)
    print("wrote", path, os.path.getsize(path), "bytes;", dict(zip(["frame", "sphere", "camera_sets", "bsdf", "lift", "gamma"], counts.tolist())),
          "sphere hits:", int(sphere_out[:, 0].sum()), "Le rows lit / dark / missed:", int((le_out[:, 2] > 0).sum()), int(((le_out[:, 0] == 1) & (le_out[:, 2] == 0)).sum()),
          int((le_out[:, 0] == 0).sum()), "scene hits on the doubled first sphere:", ties)


if __name__ == "__main__":
    main()
