"""Outputs of the REFERENCE itself (tests/golden/ref/<fixture>.txt, produced by `julia tools/reference_golden.jl` on a
machine with Julia 0.6 + Celeste.jl) against the C oracle and -- with -m gpu -- the HIP engine.

No such file can be produced in the build image (no Julia; DESIGN.md section 2: parity unpinned): the tests skip until
someone commits one.  What IS checked here without Julia: the raw exports the Julia script reads are exactly the committed
golden inputs, and the text format the script writes round-trips."""
import os

import numpy as np
import pytest

import golden_util as gu

HERE = os.path.dirname(os.path.abspath(__file__))
RAW = os.path.join(HERE, "golden", "raw")
REF = os.path.join(HERE, "golden", "ref")
RTOL = 1e-8          # BASELINE.json: "ELBO value/gradient within 1e-8 relative of the reference"


def read_raw(name):
    out = {}
    blob = open(os.path.join(RAW, name + ".bin"), "rb").read()
    for line in open(os.path.join(RAW, name + ".txt")):
        tok = line.split()
        nd = int(tok[2])
        dims = tuple(int(x) for x in tok[3:3 + nd])
        off = int(tok[3 + nd])
        dt = np.dtype(tok[1]).newbyteorder("<")
        n = int(np.prod(dims))
        out[tok[0]] = np.frombuffer(blob, dtype=dt, count=n, offset=off).reshape(dims, order="F")
    return out


def read_ref(name):
    """{s: (v, d[44], h[44, 44], neighbours)} from the Julia script's output"""
    res = {}
    for line in open(os.path.join(REF, name + ".txt")):
        if line.startswith("#") or not line.strip():
            continue
        tok = line.split()
        s = int(tok[1])
        rec = res.setdefault(s, {})
        vals = tok[2:]
        if tok[0] == "v":
            rec["v"] = float(vals[0])
        elif tok[0] == "d":
            rec["d"] = np.array(vals, dtype=np.float64)
        elif tok[0] == "h":
            rec["h"] = np.array(vals, dtype=np.float64).reshape(44, 44, order="F")
        elif tok[0] == "n":
            rec["n"] = [int(x) for x in vals]
    return res


RAW_NAMES = sorted(f[:-4] for f in os.listdir(RAW) if f.endswith(".txt")) if os.path.isdir(RAW) else []
REF_NAMES = sorted(f[:-4] for f in os.listdir(REF) if f.endswith(".txt")) if os.path.isdir(REF) else []


@pytest.mark.parametrize("name", RAW_NAMES)
def test_raw_export_is_the_committed_fixture(name):
    """what the Julia script reads == what the oracle and the HIP engine are tested on"""
    z = np.load(gu.path(name))
    f = gu.arrays_to_field(z)
    a = read_raw(name)
    assert int(a["n_images"][0]) == len(f.images) and int(a["n_sources"][0]) == len(f.catalog)
    for n, im in enumerate(f.images, 1):
        assert np.array_equal(a["pixels_%d" % n], im.pixels, equal_nan=True) and np.array_equal(a["sky_%d" % n], im.sky)
        assert np.array_equal(a["nelec_per_nmgy_%d" % n], im.nelec_per_nmgy) and int(a["band_%d" % n][0]) == im.b
        assert np.array_equal(a["psf_%d" % n], im.psf) and np.array_equal(a["psf_stamp_%d" % n], im.psfmap.stamp)
    assert np.array_equal(a["vp"], f.vp) and np.array_equal(a["pos"], np.array([c.pos for c in f.catalog]))


def test_reference_output_format_round_trips(tmp_path, monkeypatch):
    rng = np.random.default_rng(0)
    v, d, h = rng.standard_normal(), rng.standard_normal(44), rng.standard_normal((44, 44))
    with open(os.path.join(str(tmp_path), "x.txt"), "w") as fh:
        fh.write("# comment\nv 0 %.17g\nd 0 %s\nh 0 %s\nn 0 2 5\n" % (v, " ".join("%.17g" % x for x in d),
                                                                     " ".join("%.17g" % x for x in h.flatten(order="F"))))
    monkeypatch.setattr("test_reference_outputs.REF", str(tmp_path))
    r = read_ref("x")[0]
    assert r["v"] == v and np.array_equal(r["d"], d) and np.array_equal(r["h"], h) and r["n"] == [2, 5]


def _compare(name, evaluate):
    z = np.load(gu.path(name))
    f = gu.arrays_to_field(z)
    ref = read_ref(name)
    assert sorted(ref) == list(range(len(f.catalog)))
    for s, r in ref.items():
        assert r["n"] == sorted(f.neighbors[s]) or sorted(r["n"]) == sorted(f.neighbors[s]), "Model.find_neighbors"
    v, d, h = evaluate(f)
    for s, r in ref.items():
        assert abs(v[s] - r["v"]) <= RTOL * abs(r["v"]), (name, s, "value")
        assert np.abs(d[s] - r["d"]).max() <= RTOL * np.abs(r["d"]).max(), (name, s, "gradient")
        hs = 0.5 * (r["h"] + r["h"].T)                 # the reference's Hessian is symmetric to rounding only (A22)
        assert np.abs(h[s] - hs).max() <= RTOL * np.abs(hs).max(), (name, s, "Hessian")


@pytest.mark.skipif(not REF_NAMES, reason="no reference outputs committed (needs Julia 0.6 + Celeste.jl: tools/reference_golden.jl)")
@pytest.mark.parametrize("name", REF_NAMES or ["none"])
def test_oracle_matches_the_reference(oracle, name):
    from celeste_jl_amd import cabi

    def evaluate(f):
        pb = cabi.Problem(f.images, f.patches, f.neighbors)
        v, d, h, cnt, st = oracle.elbo_batch(pb, f.vp, list(range(len(f.catalog))), 7, n_threads=1)
        assert (st == 0).all()
        return v, d, h
    _compare(name, evaluate)


@pytest.mark.gpu
@pytest.mark.skipif(not REF_NAMES, reason="no reference outputs committed (needs Julia 0.6 + Celeste.jl: tools/reference_golden.jl)")
@pytest.mark.parametrize("name", REF_NAMES or ["none"])
def test_hip_engine_matches_the_reference(name):
    import celeste_jl_amd as cel

    def evaluate(f):
        ctx = cel.FieldContext(f.images, f.patches, f.neighbors)
        v, d, h, cnt, st = ctx.eval_batch(f.vp, list(range(len(f.catalog))), 7)
        assert (st == 0).all()
        return v, d, h
    _compare(name, evaluate)
