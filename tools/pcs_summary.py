#!/usr/bin/env python
"""rocprofv3 PC-sampling CSV -> samples per kernel and per source line (Instruction_Comment carries file:line when the code
object has line tables).  usage: python tools/pcs_summary.py <rocprofv3 output dir>"""
import glob
import os
import re
import sys
from collections import Counter

import pandas as pd

d = sys.argv[1]
files = glob.glob(os.path.join(d, "**", "*pc_sampling*.csv"), recursive=True)
if not files:
    print("no pc sampling csv under", d)
    sys.exit(0)
df = pd.concat([pd.read_csv(f) for f in files])
print("columns:", list(df.columns))
print("samples:", len(df))
kt = glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True)
names = {}
if kt:
    k = pd.read_csv(kt[0])
    names = dict(zip(k["Dispatch_Id"], k["Kernel_Name"]))
did = "Dispatch_Id" if "Dispatch_Id" in df.columns else None
if did:
    df["kernel"] = df[did].map(names).fillna("?").str.replace(r"\(.*", "", regex=True).str.replace("void dsq::", "")
    top = df["kernel"].value_counts()
    print("\n== samples per kernel"); print(top.head(12).to_string())
com = "Instruction_Comment" if "Instruction_Comment" in df.columns else None
ins = "Instruction" if "Instruction" in df.columns else None
for kern in ("fit_disp_kernel<4, false, true, 0>", "fit_beta_cell_kernel<4, false>"):
    sub = df[df["kernel"] == kern] if did else df
    if not len(sub):
        continue
    print("\n==== %s: %d samples" % (kern, len(sub)))
    if ins:
        op = sub[ins].astype(str).str.split().str[0]
        cls = op.map(lambda o: "f64" if re.search(r"_f64", o) else ("ds" if o.startswith("ds_") else ("mem" if re.match(r"(global|flat|buffer|scratch)_", o) else ("salu" if o.startswith("s_") else ("lane" if re.search(r"readlane|writelane|dpp|permlane|swap|readfirstlane", o) else ("cndmask" if "cndmask" in o else ("cmp" if "_cmp" in o else ("mov" if "_mov" in o else "other_valu"))))))))
        print("-- by instruction class"); print((cls.value_counts() / len(sub)).round(3).to_string())
        print("-- top opcodes"); print((op.value_counts() / len(sub)).round(3).head(25).to_string())
    if com:
        loc = sub[com].astype(str).str.extract(r"([\w\.]+\.h(?:pp|ip)?:\d+)")[0].fillna("?")
        print("-- top source lines"); print((loc.value_counts() / len(sub)).round(4).head(40).to_string())
        fil = loc.str.replace(r":\d+", "", regex=True)
        print("-- by file"); print((fil.value_counts() / len(sub)).round(3).to_string())
