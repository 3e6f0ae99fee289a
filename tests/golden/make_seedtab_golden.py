#!/usr/bin/env python3
"""Dump the SEEDPATTERN3 tables of the reference header as DATA
(tests/golden/seedpattern3.json).  Run in the build container only."""
import json
import os
import re
import sys


def parse(path):
    src = open(path).read()
    blk = src[src.index("#ifdef SEEDPATTERN3"):]

    def body(name):
        m = re.search(name + r"\[[^\]]*\](?:\[[^\]]*\])?\s*=\s*\{(.*?)\};", blk, re.S)
        return re.sub(r"/\*.*?\*/", "", m.group(1))

    care = [int(x) for x in re.findall(r"\d+", body("F2CAREDPOSITION"))]
    rows = re.findall(r"\{([^{}]*)\}", body("F2NOCAREDPOSITION"))
    nocare = []
    for r in rows:
        v = [int(x) for x in re.findall(r"\d+", r)]
        nocare.append(v + [0] * (150 - len(v)))  # C++ zero-fills the rest of each row
    return {"F2CAREDPOSITION": care, "F2NOCAREDPOSITION": nocare}


if __name__ == "__main__":
    hdr = sys.argv[1] if len(sys.argv) > 1 else "/root/reference/src/walt/seedpattern.hpp"
    out = os.path.join(os.path.dirname(os.path.abspath(__file__)), "seedpattern3.json")
    with open(out, "w") as f:
        json.dump(parse(hdr), f)
    print("wrote", out)
