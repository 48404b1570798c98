#!/usr/bin/env python3
"""Instruction-category trace of one kernel from `hipcc -S --cuda-device-only` output.
M mfma, D ds_read, W ds_write, G LDS-DMA, g global load, S global store, |B| s_barrier, [..] s_waitcnt, <..> branches,
'.' anything else (runs of >= 8 are shown as .{n}).   usage: asm_seq.py FILE.s NAME_REGEX [max_chars]"""
import re
import sys


def main():
    path, pat = sys.argv[1], sys.argv[2]
    lim = int(sys.argv[3]) if len(sys.argv) > 3 else 4000
    text = open(path).read().split("\n")
    starts = [(i, m.group(1)) for i, l in enumerate(text) if (m := re.match(r"^(_Z\w+):", l))]
    for n, (i, name) in enumerate(starts):
        if not re.search(pat, name):
            continue
        end = starts[n + 1][0] if n + 1 < len(starts) else len(text)
        out = []
        tot = {}
        for l in text[i:end]:
            if re.match(r"^\.LBB\w+:", l):
                out.append("\n" + l.split(":")[0] + ": ")
                continue
            if not l.startswith("\t"):
                if "NumVgprs" in l or "ScratchSize" in l or "Occupancy" in l or "NumAgprs" in l:
                    out.append("\n" + l.strip())
                continue
            t = l.strip().split(";")[0].strip()
            if not t or t.startswith("."):
                continue
            if t.startswith("v_mfma"): c = "M"
            elif t.startswith("ds_read"): c = "D"
            elif t.startswith("ds_write"): c = "W"
            elif t.startswith(("global_load_lds", "buffer_load")) and "lds" in t: c = "G"
            elif t.startswith(("global_load", "flat_load", "buffer_load")): c = "g"
            elif t.startswith(("global_store", "flat_store", "buffer_store")): c = "S"
            elif t.startswith("scratch_"): c = "X"
            elif t.startswith("s_waitcnt"): c = "[" + t.replace("s_waitcnt ", "") + "]"
            elif t.startswith("s_barrier"): c = "|B|"
            elif t.startswith(("s_cbranch", "s_branch")): c = "<" + t.split()[0][2:] + " " + t.split()[-1] + ">"
            elif t.startswith("s_endpgm"): c = "END"
            else: c = "."
            tot[c[0]] = tot.get(c[0], 0) + 1
            out.append(c)
        s = re.sub(r"\.{8,}", lambda m: ".{%d}" % len(m.group()), "".join(out))
        print("==", name, {k: v for k, v in tot.items() if k in "MDWGgSX|"})
        print(s[:lim])
        if len(s) > lim:
            print("... (%d more chars)" % (len(s) - lim))
            print(s[-600:])


if __name__ == "__main__":
    main()
