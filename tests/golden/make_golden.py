#!/usr/bin/env python3
"""Regenerate tests/golden/ from the reference (run in the build container only).

Inputs : data files the reference holds as its de-facto fixtures
         (/root/reference/regex_GPU_PHF/{experimentpattern,experimentinput,1M,xaa..xad,bytefile/*}).
Outputs: * tests/golden/data/            copies of the small DATA files (patterns / inputs; no source code);
                                         `1M` is stored as its 402-byte period (`paragraph402`) and
                                         `bytefile_1000000byte` gzip-compressed
         * tests/golden/out/*.txt        small GPU_match_result.txt files, verbatim
         * tests/golden/fingerprints.json  (lines, bytes, md5) of every golden output + table statistics

Every output is produced by oracle/_ref/libpfacref.so = the reference's REAL host sources
(create_table_reorder.c, phf.c) compiled where they lie, driving the oracle's restatement of the
CUDA kernel + merge + emitter (oracle/ref_harness.cc: ref_run).  The reference's CUDA build cannot
run here (no nvcc / NVIDIA device), see DESIGN.md.
"""
import ctypes as C
import gzip
import hashlib
import json
import os
import shutil
import sys
import tempfile

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference/regex_GPU_PHF"

CASES = [  # (name, pattern, input, streams, width, keep_verbatim)
    ("exp_x_expinput_s1_w256", "experimentpattern", "experimentinput", 1, 256, True),
    ("exp_x_expinput_s2_w1024", "experimentpattern", "experimentinput", 2, 1024, False),
    ("exp_x_1M_s1_w256", "experimentpattern", "1M", 1, 256, False),          # BASELINE config 1
    ("xaa_x_1M_s1_w256", "xaa", "1M", 1, 256, False),
    ("all_x_1M_s1_w256", "xaa+xab+xac+xad", "1M", 1, 256, False),
    ("all_x_1M_s2_w256", "xaa+xab+xac+xad", "1M", 2, 256, False),
    ("all_x_1M_s1_w4096", "xaa+xab+xac+xad", "1M", 1, 4096, False),
    ("all_x_1M_s3_w1024", "xaa+xab+xac+xad", "1M", 3, 1024, False),
    ("b10000_x_1M_s1_w256", "bytefile/10000byte", "1M", 1, 256, True),
    ("b10000_x_b1000000_s1_w256", "bytefile/10000byte", "bytefile/1000000byte", 1, 256, True),
    ("b100000_x_b1000000_s1_w256", "bytefile/100000byte", "bytefile/1000000byte", 1, 256, False),
]


def main():
    if not os.path.isdir(REF):
        sys.exit("reference not present: goldens can only be regenerated in the build container")
    lib = os.path.join(REPO, "oracle", "_ref", "libpfacref.so")
    ref = C.CDLL(lib)
    ref.ref_run.restype = C.c_longlong
    ref.ref_run.argtypes = [C.c_char_p, C.c_int, C.c_int, C.c_char_p, C.c_char_p]
    data = os.path.join(HERE, "data")
    out = os.path.join(HERE, "out")
    os.makedirs(data, exist_ok=True)
    os.makedirs(out, exist_ok=True)
    # --- data copies
    for f in ("experimentpattern", "experimentinput", "xaa", "xab", "xac", "xad"):
        shutil.copyfile(os.path.join(REF, f), os.path.join(data, f))
    for f in ("10000byte", "100000byte"):
        shutil.copyfile(os.path.join(REF, "bytefile", f), os.path.join(data, "bytefile_" + f))
    with open(os.path.join(REF, "bytefile", "1000000byte"), "rb") as fi, \
            gzip.GzipFile(os.path.join(data, "bytefile_1000000byte.gz"), "wb", mtime=0) as fo:
        fo.write(fi.read())
    one_m = open(os.path.join(REF, "1M"), "rb").read()
    para = one_m[:402]
    assert (para * (len(one_m) // 402 + 1))[: len(one_m)] == one_m, "1M is not the 402-byte paragraph tiled"
    open(os.path.join(data, "paragraph402"), "wb").write(para)
    fp = {"inputs": {"1M": {"bytes": len(one_m), "md5": hashlib.md5(one_m).hexdigest(), "period": 402}}, "cases": {}}
    tmp = tempfile.mkdtemp()
    allpat = os.path.join(tmp, "all.pat")
    with open(allpat, "wb") as f:
        for p in ("xaa", "xab", "xac", "xad"):
            f.write(open(os.path.join(REF, p), "rb").read())
    fp["inputs"]["xaa+xab+xac+xad"] = {"md5": hashlib.md5(open(allpat, "rb").read()).hexdigest()}
    for name, pat, inp, streams, width, keep in CASES:
        ppath = allpat if "+" in pat else os.path.join(REF, pat)
        opath = os.path.join(tmp, name + ".txt")
        n = ref.ref_run(ppath.encode(), streams, width, os.path.join(REF, inp).encode(), opath.encode())
        assert n >= 0, (name, n)
        blob = open(opath, "rb").read()
        fp["cases"][name] = {"pattern": pat, "input": inp, "streams": streams, "width": width, "lines": int(n),
                             "bytes": len(blob), "md5": hashlib.md5(blob).hexdigest(), "verbatim": bool(keep)}
        if keep:
            shutil.copyfile(opath, os.path.join(out, name + ".txt"))
        print(name, n, fp["cases"][name]["md5"], flush=True)
    json.dump(fp, open(os.path.join(HERE, "fingerprints.json"), "w"), indent=1, sort_keys=True)
    shutil.rmtree(tmp)


if __name__ == "__main__":
    main()
