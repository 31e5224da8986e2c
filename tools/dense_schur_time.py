"""the two forms of the Schur complement side by side on scenes with dense visibility: usage: python tools/dense_schur_time.py [cams pts] ...
(default: 120 x 8000, 300 x 8000, 1000 x 20000); prints ms per launch of the pair plan (where the engine built one) and of the dense product"""
import importlib, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
st = importlib.import_module("slam-tricks_amd"); scenes = importlib.import_module("slam-tricks_amd.scenes")
a = [int(x) for x in sys.argv[1:]]
cfgs = list(zip(a[0::2], a[1::2])) or [(120, 8000), (300, 8000), (1000, 20000)]
for nc, npt in cfgs:
    t0 = time.time()
    s = scenes.st20_scene(n_cams=nc, n_pts=npt, seed=3, pos_noise=0.1, ang_noise_deg=1.5, pix_noise=1e-3, half_w=3.0, half_h=3.0, retriangulate=False)
    k = np.bincount(s["obs_pt"], minlength=len(s["pts0"]))
    pairs = int((k * (k + 1) // 2).sum())
    t1 = time.time()
    e = st.BAEngine(s["cams0"], s["pts0"], s["obs_cam"], s["obs_pt"], s["obs_feat"], s["cam_fixed"])
    t2 = time.time()
    auto = e.schur_mode()
    out = {}
    for mode, name in ((e.SCHUR_PAIRS, "pairs"), (e.SCHUR_DENSE, "dense")):
        try:
            e.set_schur_mode(mode)
        except st.StbaError as ex:
            out[name] = None
            continue
        out[name] = e.time_schur(5)[0]
    e.set_schur_mode(auto)
    tl = time.time(); summ, tr = e.lm_iterations(5); tl = (time.time() - tl) / 5
    n = 6 * nc
    flops = n * n * 3.0 * len(s["pts0"])          # lower triangle of Y Y^T: n^2 / 2 entries x K x 2
    d = out["dense"]
    print(f"{nc} cams x {len(s['pts0'])} landmarks, {len(s['obs_cam'])} observations ({len(s['obs_cam']) / (nc * len(s['pts0'])):.2f} visible), {pairs:.3g} pairs; "
          f"auto = {'dense' if auto == 2 else 'pairs'}; pairs {out['pairs'] if out['pairs'] is None else round(out['pairs'], 3)} ms, dense {d:.3f} ms "
          f"({flops / d / 1e9:.1f} TFLOP/s of the product); one LM iteration {tl * 1e3:.2f} ms; scene {t1 - t0:.1f} s, create {t2 - t1:.1f} s", flush=True)
