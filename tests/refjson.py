"""Replay a run's per-read outcome into the REAL reference Stats / FilterResult / JsonReporter
objects (oracle/_ref/ref_harness) and let the reference write fastplong.json."""
import re

import numpy as np

from fastplong_amd import abi


STAMP = re.compile(rb"\d{4}-\d\d-\d\d      \d\d:\d\d:\d\d")  # HtmlReporter::getCurrentSystemTime


def reference_json(ref, path, cfg, seq, qual, off, res, counters, c, threads, is_rna=False, frags=None, regs=None,
                   html=None, title="fastplong report"):
    """replay the oracle's per-read outcome into the reference's own Stats / FilterResult objects
    (pre reads, passing fragments, filter codes, adapter keys, polyX) and let the reference write
    the JSON (and, with `html`, the HTML report through the real HtmlReporter)."""
    s = ref.s
    v = abi.CountersView(counters, c, cfg.n_adapters)
    lines = ["J_BEGIN %d %d %d %d %d %d %s %s" % (threads, 0, int(is_rna), cfg.opt.adapter_enabled, cfg.opt.polyx,
                                               cfg.opt.complexity_filter, s(cfg.start), s(cfg.end))]
    ads = cfg.adapter_list()
    first = True
    for i in range(len(off) - 1):
        a, b = int(off[i]), int(off[i + 1])
        rs, rq = seq[a:b].tobytes().decode("latin-1"), qual[a:b].tobytes().decode("latin-1")
        lines.append("J_PRE %s %s" % (s(rs), s(rq)))
        if first:
            # aggregate-only events can go to any worker: FilterResult::merge sums them
            for ai in range(cfg.n_adapters):
                for side in range(2):
                    for k in np.nonzero(v.key_hist[ai, side])[0]:
                        key = ads[ai][len(ads[ai]) - k:] if side == 0 else ads[ai][:k]
                        lines += ["J_AD %s" % s(key)] * int(v.key_hist[ai, side, k])
            for bb in range(4):
                if v.polyx_reads[bb]:
                    lines.append("J_PXT %d %d" % (bb, int(v.polyx_bases[bb])))
                    lines += ["J_PXT %d 0" % bb] * (int(v.polyx_reads[bb]) - 1)
            if v.adapter_reads:
                lines.append("J_ART %d" % int(v.adapter_bases))
                lines += ["J_ART 0"] * (int(v.adapter_reads) - 1)
            first = False
        r = res[i]
        if frags is not None:  # --break / --mask: the output reads come from the fragment list
            for fr in frags[frags["read"] == i]:
                lines.append("J_FR %d" % fr["code"])
                if fr["code"] == 0:
                    fa, fb = int(fr["start"]), int(fr["start"]) + int(fr["len"])
                    sb = bytearray(rs[fa:fb].encode("latin-1"))
                    for g in regs[fr["region_first"]:fr["region_first"] + fr["region_count"]]:
                        x = int(g["start"]) - fa
                        sb[x:x + int(g["len"])] = b"N" * int(g["len"])
                    lines.append("J_POST %s %s" % (s(bytes(sb)), s(rq[fa:fb])))
            continue
        for f in range(r["n_frag"]):
            lines.append("J_FR %d" % r["code"][f])
            if r["code"][f] == 0:
                fa = int(r["frag_start"][f])
                fb = fa + int(r["frag_len"][f])
                lines.append("J_POST %s %s" % (s(rs[fa:fb]), s(rq[fa:fb])))
    if html is None:
        lines.append("J_END %s" % s(path))
    else:
        lines.append("J_ENDH %d %d %s %s %s" % (cfg.opt.length_filter, cfg.opt.max_length, s(path), s(html), "=" + title))  # the title may hold spaces: last field
    out = ref.run(lines)
    assert out.strip().endswith("OK"), out[-200:]


