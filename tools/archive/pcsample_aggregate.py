"""Aggregate a rocprofv3 PC-sampling CSV (pc_sampling_{stochastic,host_trap}.csv) on the GPU box into a summary small enough to travel:
samples per kernel, per instruction (top N) and — stochastic sampling — per stall reason / instruction type.  Usage: pcsample_aggregate.py DIR OUT.txt"""
import collections
import csv
import glob
import os
import sys

csv.field_size_limit(1 << 30)


def main(d, out):
    files = sorted(glob.glob(os.path.join(d, "**", "*pc_sampling*.csv"), recursive=True))
    kern = {}
    for kf in glob.glob(os.path.join(d, "**", "*kernel_trace.csv"), recursive=True):
        with open(kf, newline="") as f:
            for row in csv.DictReader(f):
                cid = row.get("Correlation_Id") or row.get("Dispatch_Id")
                kern[cid] = row.get("Kernel_Name", "?")[:60]
    with open(out, "w") as o:
        o.write("files: %s\n" % files)
        for fn in files:
            with open(fn, newline="") as f:
                rd = csv.DictReader(f)
                cols = rd.fieldnames
                o.write("\n== %s\ncolumns: %s\n" % (fn, cols))
                by_inst = collections.Counter(); by_kernel = collections.Counter(); extra = {c: collections.Counter() for c in cols if c not in ("Sample_Timestamp", "Exec_Mask", "Dispatch_Id", "Instruction", "Instruction_Comment", "Correlation_Id")}
                stall_by_inst = collections.defaultdict(collections.Counter)
                n = 0
                first = []
                for row in rd:
                    n += 1
                    if len(first) < 5:
                        first.append(dict(row))
                    k = kern.get(row.get("Correlation_Id"), row.get("Correlation_Id", "?"))
                    by_kernel[k] += 1
                    key = (k, row.get("Instruction", "?"), row.get("Instruction_Comment", ""))
                    by_inst[key] += 1
                    for c in extra:
                        extra[c][row.get(c)] += 1
                    for c in ("Stall_Reason", "Wave_Issued_Instruction", "Instruction_Type", "Reason_Not_Issued"):
                        if c in row:
                            stall_by_inst[key][c + "=" + str(row[c])] += 1
                o.write("samples: %d\nfirst rows: %s\n" % (n, first))
                o.write("\n-- per kernel\n")
                for k, c in by_kernel.most_common(20):
                    o.write("%9d  %s\n" % (c, k))
                for c, cnt in extra.items():
                    if 1 < len(cnt) <= 64:
                        o.write("\n-- %s\n" % c)
                        for v, x in cnt.most_common():
                            o.write("%9d  %s\n" % (x, v))
                o.write("\n-- per instruction (top 400)\n")
                for key, c in by_inst.most_common(400):
                    s = stall_by_inst.get(key)
                    o.write("%8d  %-28s %-70s %s | %s\n" % (c, key[0][:28], key[1][:70], key[2][-60:], " ".join("%s:%d" % kv for kv in s.most_common(6)) if s else ""))


if __name__ == "__main__":
    main(sys.argv[1], sys.argv[2])
