#!/usr/bin/env python3
"""Extract the NUMBERS held by the reference's own Poseidon test fixtures into JSON.

Run in the build container only (reads /root/reference, which does not exist on the GPU box):
    python tests/golden/make_ref_fixtures.py
Sources (data files `include!`d by the reference's tests, plugins/arkworks/src/poseidon/test.rs:58-65,
:477-497, and the inline KAT at openzl-tutorials/src/poseidon.rs:383-400):
  plugins/arkworks/src/poseidon/mds_hardcoded_tests/width{2..12}      Cauchy MDS over BLS12-381 Fr
  plugins/arkworks/src/poseidon/parameters_hardcoded_test/lfsr_values   189 round constants (255,3,8,55)
  plugins/arkworks/src/poseidon/permutation_hardcoded_test/width3       permutation([3,1,2])
Only decimal field values are kept (inputs/expected outputs); no reference source text is copied.
"""
import json, os, re

REF = "/root/reference/plugins/arkworks/src/poseidon"
NUM = re.compile(r'field_new!\(Fr,\s*"(\d+)"\)')


def numbers(path):
    with open(path) as f:
        return NUM.findall(f.read())


def main():
    out = {"field": "bls12_381_fr", "mds": {}, "source": "openzklib/openzl plugins/arkworks/src/poseidon/*_hardcoded_test*"}
    for t in range(2, 13):
        flat = numbers(f"{REF}/mds_hardcoded_tests/width{t}")
        assert len(flat) == t * t, (t, len(flat))
        out["mds"][str(t)] = [flat[i * t:(i + 1) * t] for i in range(t)]
    out["lfsr_values"] = numbers(f"{REF}/parameters_hardcoded_test/lfsr_values")
    assert len(out["lfsr_values"]) == 189
    out["permutation_width3"] = {"input": ["3", "1", "2"], "output": numbers(f"{REF}/permutation_hardcoded_test/width3")}
    assert len(out["permutation_width3"]["output"]) == 3
    # the same three numbers are inlined in the tutorial's KAT test
    with open("/root/reference/openzl-tutorials/src/poseidon.rs") as f:
        tut = re.findall(r'"(\d{60,})"', f.read())
    assert tut == out["permutation_width3"]["output"], tut
    dst = os.path.join(os.path.dirname(os.path.abspath(__file__)), "ref_poseidon_fixtures.json")
    with open(dst, "w") as f:
        json.dump(out, f, indent=0)
    print("wrote", dst)


if __name__ == "__main__":
    main()
