"""One-off differential campaign: the two randomised GPU-vs-oracle tests of tests/test_gpu_parity.py with many more seeds
than the suite runs, the extreme-depth scenes and the GOURAUD / PHONG soups
(python profiles/fuzz_campaign.py [n_fuzz] [n_adversarial] [n_depth] [n_gouraud] [n_phong] [n_large])."""
import sys, os, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_parity as T
nf = int(sys.argv[1]) if len(sys.argv) > 1 else 120
na = int(sys.argv[2]) if len(sys.argv) > 2 else 60
nd = int(sys.argv[3]) if len(sys.argv) > 3 else 40
ng = int(sys.argv[4]) if len(sys.argv) > 4 else 0
np_ = int(sys.argv[5]) if len(sys.argv) > 5 else 0
nl = int(sys.argv[6]) if len(sys.argv) > 6 else 0
t0 = time.time(); bad = []
for s in range(12, 12 + nf):
    try: T.test_fuzz_flat_scenes_against_oracle(s)
    except AssertionError as e: bad.append(("fuzz", s, str(e)[:80]))
    if s % 100 == 0: print(f"fuzz seed {s}: {len(bad)} mismatching so far, {time.time() - t0:.0f} s", flush=True)
for s in range(6, 6 + na):
    try: T.test_block_masks_on_adversarial_shapes(s)
    except AssertionError as e: bad.append(("adversarial", s, str(e)[:80]))
    if s % 100 == 0: print(f"adversarial seed {s}: {len(bad)} mismatching so far, {time.time() - t0:.0f} s", flush=True)
for s in range(6, 6 + nd):
    try: T.test_depth_plane_early_test_on_extreme_depths(s)
    except AssertionError as e: bad.append(("depth", s, str(e)[:80]))
    if s % 20 == 0: print(f"depth seed {s}: {len(bad)} mismatching so far, {time.time() - t0:.0f} s", flush=True)
for s in range(4, 4 + ng):
    try: T.test_fuzz_gouraud_scenes_against_oracle(s)
    except AssertionError as e: bad.append(("gouraud", s, str(e)[:80]))
for s in range(4, 4 + np_):
    try: T.test_fuzz_phong_soup_against_oracle(s)
    except AssertionError as e: bad.append(("phong", s, str(e)[:80]))
for s in range(4, 4 + nl):
    try: T.test_depth_bound_in_the_pair_on_large_triangles(s)
    except AssertionError as e: bad.append(("large", s, str(e)[:80]))
if nl: print(f"{nl} scenes of large triangles done, {len(bad)} mismatching so far, {time.time() - t0:.0f} s", flush=True)
if ng or np_: print(f"{ng} gouraud + {np_} phong soups done, {len(bad)} mismatching so far, {time.time() - t0:.0f} s", flush=True)
print(f"{nf} fuzz + {na} adversarial + {nd} extreme-depth scenes in {time.time() - t0:.0f} s: {len(bad)} mismatching", bad[:5])
