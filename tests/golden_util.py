"""Helpers shared by the parity tests: load the reference-generated golden
vectors (tests/golden/*.npz, made by tests/golden/make_golden.py) and compare
tensors against them."""
import os

import numpy as np

from oracle import numpy_oracle as O

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
FIXTURES = ["ref_b8_s3.npz", "ref_b64_s2.npz"]
N_SAMPLE = 256


def _hash_str(s):
    h = 2166136261
    for ch in s.encode():
        h = ((h ^ ch) * 16777619) & 0xFFFFFFFF
    return h


def sample_index(key, numel):
    rng = np.random.RandomState(abs(_hash_str(key)) % (2 ** 31))
    return rng.randint(0, numel, size=N_SAMPLE).astype(np.int64)


class Golden:
    def __init__(self, fname):
        self.z = np.load(os.path.join(GOLDEN_DIR, fname))
        self.batch = int(self.z["meta/batch"])
        self.steps = int(self.z["meta/steps"])
        self.seed_init = int(self.z["meta/seed_init"])
        self.seed_data = int(self.z["meta/seed_data"])

    def init_state(self):
        """Regenerate the initial state and verify it is the one the reference
        was loaded with (checksums recorded in the fixture)."""
        st = O.init_state(self.seed_init, 2, 1024)
        for k, v in st.items():
            assert np.float64(v.reshape(-1).astype(np.float64).sum()) == self.z["init/%s/sum" % k], k
            assert np.array_equal(v.reshape(-1)[:4], self.z["init/%s/head" % k]), k
        return st

    def batch_xy(self, s):
        x, t = O.synthetic_batch(self.seed_data + s, self.batch)
        assert np.array_equal(x, self.z["step%d/x" % s])
        assert np.array_equal(t, self.z["step%d/t" % s])
        return x, t

    def masks(self, s):
        return [np.unpackbits(self.z["step%d/mask%d" % (s, i)], axis=1)[:, :1024].astype(np.uint8)
                for i in range(5)]

    def scalar(self, name):
        return float(self.z[name])

    def arr(self, name):
        return self.z[name]

    def _ref_at_compare_positions(self, prefix, key):
        full_name = "%s/%s/full" % (prefix, key)
        if full_name in self.z.files:
            return self.z[full_name].reshape(-1).astype(np.float64)
        return self.z["%s/%s/sample" % (prefix, key)].astype(np.float64)

    def adam_atol(self, s, key, rtol, numel):
        """Per-element absolute slack for a parameter after Adam step ``s``.
        Adam's update lr*m_hat/(sqrt(v_hat)+eps) is sign-like, hence
        ill-conditioned where |g| is tiny: an error dg in the gradient moves the
        parameter by up to ~2*lr*dg/(sqrt(v_hat)+eps) (capped at 2*lr).  dg is the
        gradient tolerance rtol*(|g|+rms_g) itself."""
        lr = float(self.z["step%d/lr" % s])
        t = s + 1
        g = np.abs(self._ref_at_compare_positions("step%d/grad_clipped" % s, key))
        v = self._ref_at_compare_positions("step%d/exp_avg_sq" % s, key)
        rms_g = float(self.z["step%d/grad_clipped/%s/norm" % (s, key)]) / np.sqrt(numel)
        dg = rtol * (g + rms_g)
        vhat = np.sqrt(np.maximum(v, 0) / (1 - 0.999 ** t))
        return 2 * lr * t * np.minimum(1.0, dg / (vhat + 1e-8))

    def compare(self, prefix, key, arr, rtol, atol_abs=0.0):
        """Compare ``arr`` with the recorded tensor ``prefix/key``.  Error is
        measured relative to the tensor's RMS (so that near-zero elements of a
        tensor with a healthy scale do not dominate):
        |a-b| <= rtol*(|b| + rms) + atol_abs.
        Returns the worst normalised error."""
        arr = np.asarray(arr)
        flat = arr.reshape(-1).astype(np.float64)
        ref_norm = float(self.z["%s/%s/norm" % (prefix, key)])
        rms = ref_norm / np.sqrt(flat.size)
        full_name = "%s/%s/full" % (prefix, key)
        if full_name in self.z.files:
            ref = self.z[full_name].reshape(-1).astype(np.float64)
            got = flat
        else:
            ref = self.z["%s/%s/sample" % (prefix, key)].astype(np.float64)
            got = flat[sample_index(key, flat.size)]
        # element-wise bound: rounding error scales with the element, the RMS term
        # keeps near-zero elements of a healthy tensor from dominating
        bound = rtol * (np.abs(ref) + rms) + atol_abs
        excess = np.abs(got - ref) - bound
        worst = int(np.argmax(excess)) if ref.size else 0
        assert ref.size == 0 or excess[worst] <= 0, \
            "%s/%s: |err|=%.3e > bound=%.3e at %d (ref %.3e, rms %.3e)" % (
                prefix, key, abs(got[worst] - ref[worst]), bound[worst], worst, ref[worst], rms)
        err = float((np.abs(got - ref) / np.maximum(bound, 1e-300)).max()) if ref.size else 0.0
        got_norm = float(np.sqrt((flat ** 2).sum()))
        assert abs(got_norm - ref_norm) <= rtol * ref_norm + float(np.max(atol_abs)) * np.sqrt(flat.size), \
            "%s/%s: norm %.6e vs %.6e" % (prefix, key, got_norm, ref_norm)
        return err


# Linear biases that feed a BatchNorm have a mathematically zero gradient
# (SURVEY.md hazard H2): the reference holds rounding noise there (|g| ~ 1e-9)
# that Adam amplifies to updates of order lr.  They are compared with an
# absolute tolerance only.
def is_prebn_bias(key):
    return key.endswith(".0.bias") and not key.startswith("decode")


def safe_masks(st0, x, masks, rounding=None, thr=1e-4):
    """Keep-masks with every element dropped whose ReLU gate is within ``thr`` of flipping in the
    fp64 oracle forward (iterated: dropping an element moves the later stages).

    A training step evaluates millions of ReLU gates; for the handful of pre-activations within
    an fp32 ulp of zero ANY two correct implementations (fp64 oracle, NumPy fp32 oracle, PyTorch
    CPU, the HIP kernels) may open the gate differently, a discrete change of the gradients that
    says nothing about correctness.  Dropping those elements (about 0.01 %) makes every gate
    decision independent of rounding, so gradients can be compared at 1e-4 instead of 1e-3."""
    masks = [np.array(m, dtype=np.uint8, copy=True) for m in masks]
    O.set_gemm_rounding(rounding)
    try:
        # (stage l is final once stages < l are: at most one pass per stage, plus the check)
        for _ in range(max(8, len(masks) + 2)):
            st = {k: v.copy() for k, v in st0.items()}
            _, cache = O.forward(st, x, masks, training=True, dtype=np.float64)
            changed = 0
            for li, c in enumerate(cache["layers"]):
                risky = (np.abs(c["y"]) < thr) & (masks[li] != 0)
                n = int(risky.sum())
                if n:
                    masks[li][risky] = 0
                    changed += n
            if not changed:
                return masks
    finally:
        O.set_gemm_rounding(None)
    raise AssertionError("safe masks did not converge")
