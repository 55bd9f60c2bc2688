#!/usr/bin/env python
"""Static instruction mix and register budget of the SHIPPED gfx950 code objects (round 5).

    python tools/isa_mix.py deseq2_amd/libdeseq2_mi355x.so 'fit_disp_kernel<4, false, true, 0>' 'fit_beta_cell_kernel<4, false>'

For every kernel whose demangled name contains one of the filters: vgpr / sgpr / spill counts and scratch from the code
object's metadata, then the instruction classes of the whole kernel and of every backward-branch region (a loop: from the
target of a backward branch to the branch) of at least --min instructions, innermost regions first.  Classes:
  f64      v_*_f64 arithmetic (add / mul / fma / fmac / rcp / rndne / ldexp / div_* ...), not compares or conversions
  mov      v_mov_b32 / v_mov_b64 / accvgpr moves (without DPP)
  cnd      v_cndmask
  lane     v_readlane / v_writelane / v_readfirstlane (SGPR spills and lane broadcasts; the last three columns split them:
           a v_readlane whose source VGPR is the target of a v_writelane somewhere in the kernel is an SGPR-spill reload)
  dpp      DPP moves and v_permlane*_swap (cross-lane butterflies)
  smem     s_load_* (the coefficient fetches of dsq_isa.hpp among them)
The dynamic counterpart is the SQ_INSTS_VALU_*_F64 pass of tools/gpu_job.sh pmc (profiles/r05_pmc_*.json)."""
import argparse
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin/"


def classify(op, text):
    if "dpp" in text or op.startswith("v_permlane"):
        return "dpp"
    if op.startswith(("v_mov_b", "v_accvgpr")):
        return "mov"
    if op.startswith(("v_readlane", "v_writelane", "v_readfirstlane")):
        return "lane"
    if op.startswith("v_cndmask"):
        return "cnd"
    if op.startswith("v_cmp"):
        return "cmp"
    if op.startswith("v_cvt"):
        return "cvt"
    if op.startswith("v_") and "_f64" in op:
        return "f64"
    if op.startswith("v_"):
        return "valu_other"
    if op.startswith("s_load") or op.startswith("s_buffer_load"):
        return "smem"
    if op.startswith("s_"):
        return "salu"
    if op.startswith("ds_"):
        return "lds"
    if op.startswith(("global_", "flat_", "scratch_", "buffer_")):
        return "vmem"
    return "other"


COLS = ["all", "VALU", "f64", "mov", "cnd", "lane", "dpp", "cmp", "cvt", "valu_other", "salu", "smem", "lds", "vmem"]


def spill_vgprs(ins):
    """VGPRs that hold spilled SGPRs: the destinations of v_writelane"""
    out = set()
    for t in ins:
        if t.startswith("v_writelane_b32"):
            m = re.match(r"v_writelane_b32\s+(v\d+)", t)
            if m:
                out.add(m.group(1))
    return out


def tally(ins, spill=frozenset()):
    c = collections.Counter()
    for t in ins:
        if t.startswith("v_readlane_b32"):
            m = re.match(r"v_readlane_b32\s+\S+,\s*(v\d+)", t)
            c["readlane_spill" if (m and m.group(1) in spill) else "readlane_bcast"] += 1
        elif t.startswith("v_writelane_b32"):
            c["writelane"] += 1
        op = t.split()[0]
        k = classify(op, t)
        c[k] += 1
        c["all"] += 1
        if op.startswith("v_"):
            c["VALU"] += 1
        if op == "v_fmac_f64_e32":
            c["fmac"] += 1
        if op == "v_mov_b64_e32" and re.search(r"v_mov_b64_e32 v\[\d+:\d+\], v\[", t):
            c["mov64vv"] += 1
    return c


def demangle(names):
    out = subprocess.run(["c++filt"], input="\n".join(names), capture_output=True, text=True).stdout.split("\n")
    return dict(zip(names, out))


def metadata(obj):
    txt = subprocess.run([LLVM + "llvm-readelf", "--notes", obj], capture_output=True, text=True).stdout
    md = {}
    for blk in re.split(r"\n\s*- \.agpr_count:", txt)[1:]:
        name = re.search(r"\.name:\s+(\S+)", blk)
        if not name:
            continue
        g = lambda k: (re.search(r"\.%s:\s+(\d+)" % k, blk) or [None, "?"])[1]
        md[name.group(1)] = dict(vgpr=g("vgpr_count"), sgpr=g("sgpr_count"), vspill=g("vgpr_spill_count"),
                                 sspill=g("sgpr_spill_count"), scratch=g("private_segment_fixed_size"),
                                 lds=g("group_segment_fixed_size"))
    return md


def kernels_of(obj):
    text = subprocess.run([LLVM + "llvm-objdump", "-d", "--mcpu=gfx950", obj], capture_output=True, text=True).stdout
    name, base, ins, addr = None, 0, [], []
    for line in text.split("\n"):
        m = re.match(r"([0-9a-f]{16}) <([^>]+)>:", line)
        if m:
            if name and ins:
                yield name, base, ins, addr
            name, base, ins, addr = m.group(2), int(m.group(1), 16), [], []
            continue
        c = line.find("// ")
        if c < 0 or name is None:
            continue
        t = line[:c].strip()
        try:
            a = int(line[c + 3:c + 15], 16)
        except ValueError:
            continue
        mt = re.search(r"<[^>]+\+0x([0-9a-f]+)>", line[c:])
        if t.startswith(("s_cbranch", "s_branch")) and mt:
            t += " @%x" % (base + int(mt.group(1), 16))
        ins.append(t); addr.append(a)
    if name and ins:
        yield name, base, ins, addr


def regions(ins, addr):
    idx = {a: i for i, a in enumerate(addr)}
    out = set()
    for k, t in enumerate(ins):
        m = re.search(r" @([0-9a-f]+)$", t)
        if m and t.startswith(("s_cbranch", "s_branch")):
            tg = int(m.group(1), 16)
            if tg <= addr[k] and tg in idx:
                out.add((idx[tg], k))
    # merge regions sharing a head (several latches of one loop): keep the widest
    best = {}
    for lo, hi in out:
        best[lo] = max(best.get(lo, hi), hi)
    return sorted(best.items(), key=lambda r: (r[1] - r[0]))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("lib")
    ap.add_argument("filters", nargs="+")
    ap.add_argument("--min", type=int, default=150)
    a = ap.parse_args()
    tmp = tempfile.mkdtemp(prefix="dsq_mix_")
    try:
        lib = os.path.join(tmp, "lib.so")
        shutil.copy(a.lib, lib)
        subprocess.run([LLVM + "llvm-objdump", "--offloading", lib], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, cwd=tmp)
        objs = sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)
        for obj in objs:
            md = metadata(obj)
            dm = demangle(list(md))
            want = {n for n in md if any(f in dm.get(n, "") for f in a.filters)}
            if not want:
                continue
            for name, base, ins, addr in kernels_of(obj):
                if name not in want:
                    continue
                m = md[name]
                print("## `%s`\n" % dm[name])
                print("vgpr %s, sgpr %s, spilled vgpr %s, spilled sgpr %s, scratch %s B/lane, static LDS %s B\n" %
                      (m["vgpr"], m["sgpr"], m["vspill"], m["sspill"], m["scratch"], m["lds"]))
                print("| region | " + " | ".join(COLS) + " | v_fmac_f64 | v_mov_b64 v,v | readlane: SGPR reload | readlane: broadcast | writelane |")
                print("|---|" + "---|" * (len(COLS) + 5))
                sp = spill_vgprs(ins)
                rows = [("whole kernel", tally(ins, sp))]
                for lo, hi in regions(ins, addr):
                    if hi - lo + 1 >= a.min:
                        rows.append(("loop +0x%x..+0x%x" % (addr[lo] - base, addr[hi] - base), tally(ins[lo:hi + 1], sp)))
                for label, c in rows:
                    print("| %s | " % label + " | ".join(str(c[k]) for k in COLS) + " | %d | %d | %d | %d | %d |" % (c["fmac"], c["mov64vv"], c["readlane_spill"], c["readlane_bcast"], c["writelane"]))
                print()
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    sys.exit(main())
