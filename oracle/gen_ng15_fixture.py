"""Test/benchmark infrastructure (not product code): turns the reference's noise dictionary
``/root/reference/noise_dicts/ng15_dict.json`` (the parameter source of BASELINE.json config 3)
into the compact per-pulsar fixture ``tests/golden/ng15_noise.json`` that travels to the GPU box.

    python oracle/gen_ng15_fixture.py

Key grammar of the dictionary (parsed ad hoc by the reference's notebook, examples/add_noise.ipynb cell 6):
``{psr}_{backend}_efac``, ``{psr}_{backend}_log10_t2equad``, ``{psr}_{backend}_log10_ecorr``,
``{psr}_red_noise_log10_A``, ``{psr}_red_noise_gamma`` and the common ``gw_log10_A``.  Values are copied bit for bit
(json round-trips float64 exactly).  One backend (J1751-2857 Rcvr1_2_GUPPI) has no ``_efac`` entry in the dictionary; it is
recorded as null and consumers use EFAC = 1, the reference's own default (white_noise.py:47).
"""
import collections
import json
import os

SRC = "/root/reference/noise_dicts/ng15_dict.json"
DST = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests", "golden", "ng15_noise.json")
SUFFIXES = ("efac", "log10_t2equad", "log10_ecorr")


def main():
    with open(SRC) as fh:
        d = json.load(fh)
    names = sorted({k.split("_")[0] for k in d if not k.startswith("gw")})
    out = collections.OrderedDict(source="noise_dicts/ng15_dict.json (bencebecsy/pta_replicator)", gw_log10_A=d["gw_log10_A"],
                                  pulsars=collections.OrderedDict())
    used = {"gw_log10_A"}
    for p in names:
        be = collections.OrderedDict()
        for k, v in d.items():
            if not k.startswith(p + "_"):
                continue
            rest = k[len(p) + 1:]
            for suf in SUFFIXES:
                if rest.endswith("_" + suf):
                    be.setdefault(rest[:-len(suf) - 1], {})[suf] = v
                    used.add(k)
        rec = collections.OrderedDict(backends=list(be))
        for suf in SUFFIXES:
            rec[suf] = [be[b].get(suf) for b in be]
        for key in ("red_noise_log10_A", "red_noise_gamma"):
            rec[key] = d.get(f"{p}_{key}")
            used.add(f"{p}_{key}")
        out["pulsars"][p] = rec
    left = set(d) - used
    assert not left, f"unparsed keys: {sorted(left)[:5]}"
    with open(DST, "w") as fh:
        json.dump(out, fh, indent=0, separators=(",", ":"))
    print(f"wrote {DST}: {len(names)} pulsars, {sum(len(r['backends']) for r in out['pulsars'].values())} backends")


if __name__ == "__main__":
    main()
