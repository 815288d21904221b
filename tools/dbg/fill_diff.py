#!/usr/bin/env python3
"""GPU box: where do the labels of the literal 301 x 301 x 561 scene differ from the oracle, by fill form and views per launch?
(the tool that located the two wrong labels per lane of an asm store whose data registers were reused too early)"""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
from plant3dvision_amd import _native as nat, scenes
from oracle import oracle_c

shape, origin, vs, views = scenes.literal_real_plant_scene(60, "plant")
want = oracle_c.carve(shape, origin, vs, views, nthreads=16)
K = np.stack([v[0] for v in views]); R = np.stack([v[1] for v in views]); t = np.stack([v[2] for v in views])
stack = np.ascontiguousarray(np.stack([m for _, _, _, m in views]))
for spec in (3, 0):  # with and without the fill ahead of the verdicts
    for vpl in (7, 0, 13):
        eng = nat.Engine(shape, origin, vs, nat.SC_MODE_CARVE)
        eng.set_option(nat.SC_OPT_SPEC_SHARE, spec)
        ptr = eng.dev_alloc(stack.nbytes); eng.dev_upload(ptr, stack)
        V, H, W = stack.shape
        for rep in range(2):
            eng.clear()
            eng.set_option(nat.SC_OPT_VIEWS_PER_LAUNCH, vpl)
            eng.process_views_device(K, R, t, ptr, V, H, W, nat.SC_MASK_U8)
            got = eng.get_values()
            bad = np.argwhere(got != want)
            print(f"spec_share {spec} vpl {vpl} rep {rep}: {len(bad)} differ", flush=True)
            for b in bad[:24]:
                i, j, k = (int(x) for x in b)
                print("   ", (i, j, k), "got", int(got[i, j, k]), "want", int(want[i, j, k]), "strip", (i, j // 16), "brick z", k // 64, "k%64", k % 64)
        eng.dev_free(ptr); eng.close()
