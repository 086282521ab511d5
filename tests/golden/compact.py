"""Storage policy of the full-size fixtures (round 6: tests/golden/ 77 MB -> see README numbers in DESIGN.md 4).

Every array is still an output of the reference itself (make_golden.py); what changes is how much of it is kept:

  * SAMPLED keys -- tensors the tests only ever COMPARE AGAINST (never feed back as an input): one element out of every run of 16
    consecutive elements of the flattened tensor, at a position drawn from a hash of (key, run index).  A [1,3,256,256] tensor keeps
    12 288 of its 196 608 values: 16 per image row, every column class (x mod 16) hit, tile borders as likely as interiors.  The
    reduction is applied to what is WRITTEN; `load()` hands the tests a `Sampled` object and `assert_close` / `err_stats` compare the
    same positions of the engine's result.  Trajectory states that are fed back as inputs (x_t, x_T, ...), the end-to-end results
    (x_T, x_edit) and one whole forward per family (ddpm_celeba.npz) stay complete.
  * the shipped DeltaBlock weights (checkpoint/*.pth of the reference: data, needed because /root/reference does not travel) are
    stored ONCE per checkpoint (`deltablock_<name>.npz`) instead of once per fixture that uses them; `load()` merges them back
    under their old `param.*` keys.

make_golden.py writes through `save()`, so regenerating a fixture from the reference reproduces the committed file bit for bit;
`python tests/golden/compact.py --rewrite` applied the policy once to the complete fixtures of round 5 (same function).
"""
import os
import sys
import zlib

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ONE_IN = 16
TAG = "#s16"

# fixture -> keys stored as samples (pure comparison targets; checked against the tests' uses)
SAMPLED = {
    "config1_celeba_smiling.npz": ["inv_first.xt_next", "inv_first.x0_t", "inv_last.x0_t", "gen999.xt_next", "gen999.x0_t", "gen512.x0_t",
                                   "gen486.xt_next", "eta153.xt_next", "eta153.x0_t", "x_edit_noise"],
    "config1_b2_celeba_smiling.npz": ["inv0.x0_t", "inv512.xt_next", "gen768.xt_next", "gen768.x0_t", "gen307.xt_next"],
    "config3_afhq_full.npz": ["inv_first.xt_next", "inv_first.x0_t", "inv_mid.xt_next", "inv_last.x0_t"],
    "config3_afhq_dog_happy.npz": ["gen999.xt_next", "gen999.x0_t", "gen435.xt_next"],
    "config4_church_gothic.npz": ["gen999.xt_next", "gen999.x0_t", "gen384.xt_next", "gen384.x0_t", "gen358.xt_next", "gen358.x0_t"],
    "ddpm_celeba.npz": ["step_gen.xt_next", "step_gen.x0_t"],          # its three whole-forward outputs stay complete
    "iddpm_afhq.npz": ["fwd_single.et", "fwd_dual.et", "fwd_dual.et_mod", "step_gen.xt_next", "step_gen.x0_t"],
    "imagenet_adm.npz": ["fwd_dual.et", "fwd_dual.et_mod"],
    "iddpm_afhq_b2.npz": ["gen768.xt_next"],
    "imagenet_adm_traj.npz": ["gen999.xt_next"],
    "imagenet_adm_step.npz": ["step.xt_next", "step.x0_t"],
}
# fixture -> the shared file its `param.*` arrays (shipped DeltaBlock weights) live in
PARAM_FILE = {
    "config1_celeba_smiling.npz": "deltablock_celeba_smiling.npz",
    "config1_b2_celeba_smiling.npz": "deltablock_celeba_smiling.npz",
    "config3_afhq_dog_happy.npz": "deltablock_afhq_dog_happy.npz",
    "config3_afhq_full.npz": "deltablock_afhq_dog_happy.npz",
    "config4_church_gothic.npz": "deltablock_church_gothic.npz",
}

_M = np.uint64(0xFFFFFFFFFFFFFFFF)


def _mix(z):
    z = (z + np.uint64(0x9E3779B97F4A7C15)) & _M
    z = ((z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)) & _M
    z = ((z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)) & _M
    return z ^ (z >> np.uint64(31))


def sample_index(n, key):
    """Flat positions kept of an n-element tensor stored under `key`: run r of 16 elements keeps r * 16 + hash(key, r) % 16."""
    runs = n // ONE_IN
    with np.errstate(over="ignore"):
        h = _mix(_mix(np.uint64(zlib.crc32(key.encode()))) ^ (np.arange(runs, dtype=np.uint64) * np.uint64(0xD1342543DE82EF95)))
    return (np.arange(runs, dtype=np.int64) * ONE_IN + (h & np.uint64(ONE_IN - 1)).astype(np.int64))


class Sampled:
    """A comparison target of which only `sample_index(numel, key)` was stored."""

    def __init__(self, key, shape, values):
        self.key, self.shape, self.values = key, tuple(int(s) for s in shape), values
        n = 1
        for s in self.shape:
            n *= s
        self.index = sample_index(n, key)
        assert self.values.shape[0] == self.index.shape[0], (key, self.values.shape, self.index.shape)

    def take(self, tensor):
        """The stored positions of a full tensor (torch or numpy) of this shape."""
        assert tuple(tensor.shape) == self.shape, f"{self.key}: shape {tuple(tensor.shape)} vs fixture {self.shape}"
        flat = tensor.reshape(-1)
        import torch
        return flat[torch.from_numpy(self.index)] if isinstance(flat, torch.Tensor) else flat[self.index]


def save(path, arrays):
    """np.savez_compressed(path, **arrays) under the policy above (arrays: name -> numpy)."""
    name = os.path.basename(path)
    arrays = dict(arrays)
    pfile = PARAM_FILE.get(name)
    if pfile:
        params = {k: arrays.pop(k) for k in [k for k in arrays if k.startswith("param.")]}
        ppath = os.path.join(os.path.dirname(path), pfile)
        if os.path.exists(ppath):
            old = np.load(ppath)
            assert set(old.files) == set(params) and all(np.array_equal(old[k], params[k]) for k in params), \
                f"{name}: its DeltaBlock weights differ from {pfile}"
        else:
            np.savez_compressed(ppath, **params)
        arrays["param_file"] = np.array(pfile)
    for k in SAMPLED.get(name, []):
        a = np.ascontiguousarray(arrays.pop(k))
        arrays[k + TAG] = a.reshape(-1)[sample_index(a.size, k)]
        arrays[k + TAG + ".shape"] = np.array(a.shape, dtype=np.int64)
    np.savez_compressed(path, **arrays)


def load(path):
    """name -> torch tensor | Sampled, with the shared DeltaBlock weights merged back under `param.*`."""
    import torch
    z = np.load(path)
    out = {}
    for k in z.files:
        if k == "param_file":
            p = np.load(os.path.join(os.path.dirname(path), str(z[k])))
            out.update({pk: torch.from_numpy(p[pk]) for pk in p.files})
        elif k.endswith(TAG + ".shape"):
            continue
        elif k.endswith(TAG):
            base = k[:-len(TAG)]
            out[base] = Sampled(base, z[k + ".shape"], torch.from_numpy(z[k]))
        else:
            out[k] = torch.from_numpy(z[k])
    return out


def rewrite(src=HERE):
    """Apply the policy to the COMPLETE fixtures under `src` (default: in place), writing the compact files next to this script."""
    before = after = 0
    for name in sorted(set(SAMPLED) | set(PARAM_FILE)):
        path = os.path.join(HERE, name)
        z = np.load(os.path.join(src, name))
        if "param_file" in z.files or any(k.endswith(TAG) for k in z.files):
            print(f"{name}: already compact")
            continue
        before += os.path.getsize(os.path.join(src, name))
        save(path, {k: z[k] for k in z.files})
        after += os.path.getsize(path)
        print(f"{name}: -> {os.path.getsize(path) / 1e6:.2f} MB")
    print(f"rewritten: {before / 1e6:.1f} MB -> {after / 1e6:.1f} MB (+ the shared deltablock_*.npz files)")


if __name__ == "__main__":
    if "--rewrite" in sys.argv:
        i = sys.argv.index("--rewrite")
        rewrite(sys.argv[i + 1] if len(sys.argv) > i + 1 else HERE)
