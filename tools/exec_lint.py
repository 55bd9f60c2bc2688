#!/usr/bin/env python
"""Static check of the COMPILER'S OUTPUT for a miscompile met in round 4 (profiles/r04_exec_remat.md): under register
pressure the allocator re-materialises a constant (v_mov_b32 vN, 0x260 ...) at the HEAD of a join block, ABOVE the
`s_or_b64 exec, exec, s[a:b]` that re-activates the lanes.  The move then runs under the mask of the region that just
ended -- empty after a loop -- and every lane reads garbage from vN afterwards (there: the class mask of an inlined
sqrt, which then returned its argument).  Nothing in the source can provoke or prevent it, so the build is checked:

    python tools/exec_lint.py deseq2_amd/libdeseq2_mi355x.so      # the shipped library: every gfx950 code object in it
    python tools/exec_lint.py file.s                               # hipcc -S --cuda-device-only output

Reports every constant move into a vector register that sits between the head of a block where lanes come back together
(the target of a forward skip on an empty mask -- s_cbranch_execz, not the forward s_cbranch_execnz that enters the
body of a masked region --, the fall-through of a loop latch; in -S output: any label) and the exec
restore that leads the block -- and, at the fall-through of a loop latch, where the mask is EMPTY, every vector write of
any kind (disassembly only).  Exit status 1 when anything is found."""
import os
import re
import shutil
import subprocess
import sys
import tempfile

OBJDUMP = "/opt/rocm/lib/llvm/bin/llvm-objdump"
SCALAR_OK = ("s_mov", "s_movk", "s_brev", "s_nop", "s_waitcnt", "v_readlane", "v_readfirstlane", "v_writelane", "s_getpc",
             "s_add_u32", "s_addc_u32")


# a constant put into a vector register: what the allocator re-materialises instead of keeping or spilling it
CONST_MOVE = re.compile(r"^(v_mov_b32_e32|v_mov_b64_e32|v_bfrev_b32_e32|v_accvgpr_write_b32)\s+\S+,\s*(-?0x[0-9a-fA-F]+|-?\d+(\.\d+)?)\s*(<.*)?$")


def scan(blocks):
    """blocks: iterable of (kernel, [instruction text ...]) with each list starting at a block head"""
    hits = []
    for kernel, where, ins in blocks:
        pending = []
        for w, t in zip(where, ins):
            op = t.split()[0]
            if op == "s_or_b64" and re.match(r"s_or_b64\s+exec,\s*exec,", t):
                hits += [(kernel, pw, pt) for pw, pt in pending]
                break
            if CONST_MOVE.match(t):
                pending.append((w, t))
            elif kernel.endswith("[loop exit]") and op.startswith("v_") and not op.startswith(("v_readlane", "v_readfirstlane", "v_writelane", "v_cmp")):
                pending.append((w, t))        # at a loop exit the mask is empty: ANY vector write in front of the restore is lost
                                              # (v_writelane / v_readlane -- SGPR spills -- do not look at the mask)
            elif not op.startswith(SCALAR_OK):
                break
    return hits


def blocks_of_asm(path):
    kernel, cur, where = None, None, None
    for ln, line in enumerate(open(path, errors="replace"), 1):
        t = line.strip()
        if not t or t.startswith((";", "//")):
            continue
        mk = re.match(r"^([A-Za-z_][\w$.]*):", t)
        if mk and not t.startswith("."):
            kernel = mk.group(1)
        if re.match(r"^\.LBB\d+_\d+:", t) or (mk and not t.startswith(".")):
            if cur:
                yield kernel, where, cur
            cur, where = [], []
            continue
        if t.startswith(".") or cur is None:
            continue
        cur.append(t); where.append("%s:%d" % (path, ln))
    if cur:
        yield kernel, where, cur


def blocks_of_disassembly(text, tag):
    """heads where the lanes of a region come back together -- the target of a forward skip on an empty mask
    (s_cbranch_execz), the fall-through of a loop's latch (a backward s_cbranch_execnz) -- with the instructions behind them"""
    kernel, base, ins, addr = None, 0, [], []

    def flush():
        if kernel is None or not ins:
            return
        idx = {a: i for i, a in enumerate(addr)}
        heads = {}
        for k, t in enumerate(ins):
            if t.startswith("s_cbranch_exec"):
                mt = re.search(r"\+0x([0-9a-f]+)>", t)
                if not mt:
                    continue
                tg = base + int(mt.group(1), 16)
                if tg > addr[k]:
                    if t.startswith("s_cbranch_execz"):
                        heads.setdefault(tg, "join")
                    # a forward s_cbranch_execnz enters the BODY of a masked region (`s_and_saveexec; s_cbranch_execnz
                    # body; s_branch join; body: v_mov ..; join: s_or exec`): writes there are the region's own, under
                    # its own non-empty mask -- e.g. the `pivot = 1.0` of a dropped column in the weighted fit_disp<4>
                elif k + 1 < len(ins) and t.startswith("s_cbranch_execnz"):
                    heads[addr[k + 1]] = "exit"       # the fall-through of a loop latch: reached with an EMPTY mask
        for tg in sorted(heads):
            i = idx.get(tg)
            if i is not None:
                chunk = range(i, min(i + 24, len(ins)))
                yield (kernel + ("  [loop exit]" if heads[tg] == "exit" else ""),
                       ["%s %s+0x%x" % (tag, kernel[:60], addr[j] - base) for j in chunk],
                       [re.sub(r"\s*<[^>]+>$", "", ins[j]) for j in chunk])

    for line in (text.split("\n") if isinstance(text, str) else text):
        line = line.rstrip("\n")
        c = line.find("// ")
        if c < 0:
            if line[:1] in "0123456789abcdef" and line.endswith(">:"):
                yield from flush()
                m = re.match(r"([0-9a-f]{16}) <([^>]+)>:", line)
                kernel, base, ins, addr = (m.group(2), int(m.group(1), 16), [], []) if m else (None, 0, [], [])
            continue
        t = line[:c].strip()
        rest = line[c + 3:]
        try:
            a = int(rest[:12], 16)
        except ValueError:
            continue
        if t.startswith("s_cbranch_exec"):
            lt = rest.find("<")
            if lt >= 0:
                t += " " + rest[lt:]
        ins.append(t); addr.append(a)
    yield from flush()


def _lint_object(path):
    # (streamed: the disassembly of the widest builds runs to gigabytes of text)
    f = os.path.basename(path)
    with subprocess.Popen([OBJDUMP, "-d", path], stdout=subprocess.PIPE, text=True, bufsize=1 << 20) as proc:
        hits = scan(blocks_of_disassembly(proc.stdout, f.split(".")[2] if f.count(".") > 2 else f))
        if proc.wait() != 0:
            raise RuntimeError("llvm-objdump failed on " + path)
    return hits


def lint_library(so):
    import multiprocessing
    hits = []
    tmp = tempfile.mkdtemp(prefix="dsq_lint_")
    try:
        lib = os.path.join(tmp, "lib.so")
        shutil.copy(so, lib)
        subprocess.run([OBJDUMP, "--offloading", lib], check=True, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
        objs = sorted((os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f), key=os.path.getsize, reverse=True)
        with multiprocessing.Pool(min(32, max(1, os.cpu_count() or 1))) as pool:
            for h in pool.imap_unordered(_lint_object, objs, chunksize=1):      # (largest first, one at a time: the widest
                hits += h                                                       #  builds take ten times the narrow ones)
        hits.sort()
        return hits, len(objs)
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


def main(argv):
    bad = 0
    for p in argv:
        if p.endswith(".so") or p.endswith(".o"):
            hits, n = lint_library(p)
            print("exec_lint: %s: %d gfx950 code objects" % (p, n))
        else:
            hits = scan(blocks_of_asm(p))
        for kernel, w, t in hits:
            bad += 1
            print("%s: %s\n    in %s" % (w, t, kernel))
    print("exec_lint: %d suspicious instruction(s)" % bad)
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main(sys.argv[1:]))
