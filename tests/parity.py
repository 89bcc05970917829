"""Shared comparison helpers for the parity tests (emulated kernels and real GPU alike)."""
import numpy as np

from fastplong_amd import abi

FIELDS = ["r1_start", "r1_len", "frag_start", "frag_len", "n_frag", "dropped", "code", "kind",
          "median_q_pre", "median_q_post"]


def canon_results(res):
    """zero the fields the ABI leaves undefined so that records compare bit for bit"""
    r = res.copy()
    r["reserved"] = 0
    for i in range(2):
        unused = r["n_frag"] <= i
        r["frag_start"][unused, i] = 0
        r["frag_len"][unused, i] = 0
        r["code"][unused, i] = 0
        r["kind"][unused, i] = 0
        notpass = unused | (r["code"][:, i] != abi.FPL_PASS_FILTER)
        r["median_q_post"][notpass, i] = 0
    d = r["dropped"] != 0
    r["r1_start"][d] = 0
    r["r1_len"][d] = 0
    return r


def assert_results_equal(got, want, seq=None, off=None):
    g, w = canon_results(got), canon_results(want)
    assert len(g) == len(w)
    for f in FIELDS:
        if not np.array_equal(g[f], w[f]):
            bad = np.nonzero((g[f] != w[f]).reshape(len(g), -1).any(axis=1))[0]
            i = int(bad[0])
            msg = "field %s differs at %d reads, first read %d: got %s want %s\n got=%s\nwant=%s" % (
                f, len(bad), i, g[f][i], w[f][i], g[i], w[i])
            if seq is not None:
                a, b = int(off[i]), int(off[i + 1])
                msg += "\nseq=%s" % seq[a:b].tobytes()[:400]
            raise AssertionError(msg)


def assert_counters_equal(got, want, c, nad):
    if np.array_equal(got, want):
        return
    g, w = abi.CountersView(got, c, nad), abi.CountersView(want, c, nad)
    for name in ("pre", "post"):
        sg, sw = getattr(g, name), getattr(w, name)
        for f in ("reads", "length_sum", "base_qual_hist", "median_hist", "median_bases", "kmer", "cyc"):
            a, b = np.asarray(getattr(sg, f)), np.asarray(getattr(sw, f))
            if not np.array_equal(a, b):
                idx = np.argwhere(a != b)[:5]
                raise AssertionError("%s.%s differs at %d cells, first %s: got %s want %s" % (
                    name, f, int((a != b).sum()), idx.tolist(), a[a != b][:5], b[a != b][:5]))
    for f in ("filter", "adapter_reads", "adapter_bases", "polyx_reads", "polyx_bases", "key_hist"):
        a, b = np.asarray(getattr(g, f)), np.asarray(getattr(w, f))
        if not np.array_equal(a, b):
            idx = np.argwhere(a != b)[:5]
            raise AssertionError("%s differs, first %s: got %s want %s" % (f, idx.tolist(), a[a != b][:5], b[a != b][:5]))
    raise AssertionError("counter buffers differ outside the named views")


def assert_fragments_equal(got_f, got_r, want_f, want_r):
    """--break / --mask outcome lists: records in (read, seq_no) order; the region list may be laid out
    differently, so regions are compared per fragment"""
    assert len(got_f) == len(want_f), (len(got_f), len(want_f))
    for name in ("read", "seq_no", "start", "len", "region_count", "break_no", "code", "kind"):
        if not np.array_equal(got_f[name], want_f[name]):
            i = int(np.nonzero(got_f[name] != want_f[name])[0][0])
            raise AssertionError("fragment field %s differs first at record %d: got %s want %s" % (name, i, got_f[i], want_f[i]))
    ok = want_f["code"] == abi.FPL_PASS_FILTER
    assert np.array_equal(got_f["median_q"][ok], want_f["median_q"][ok])
    for i in range(len(want_f)):
        a, b = int(got_f["region_first"][i]), int(want_f["region_first"][i])
        n = int(want_f["region_count"][i])
        assert np.array_equal(got_r[a:a + n], want_r[b:b + n]), ("regions of fragment", i, got_r[a:a + n], want_r[b:b + n])
