"""Basic blocks of one kernel in the device assembly (hipcc -S --cuda-device-only): label, first line, vector / scalar /
memory instruction counts and the branches that end it -- for walking a kernel's hot path by hand.
usage: asm_blocks.py file.s mangled_kernel_name [dump.s]"""
import re, sys
asm = open(sys.argv[1]).read()
sym = sys.argv[2]
i = asm.index(sym + ":")
lines = asm[i:asm.index(".end_amdhsa_kernel", i)].split("\n")
if len(sys.argv) > 3:
    open(sys.argv[3], "w").write("\n".join(lines))
blocks, cur = [], ["entry", 0, 0, 0, [], 0]
for k, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m:
        blocks.append(cur)
        cur = [m.group(1), 0, 0, 0, [], k]
        continue
    m = re.match(r"^\s+([a-z_0-9]+)\s*(.*)", l)
    if not m:
        continue
    op = m.group(1)
    if op.startswith("v_"):
        cur[1] += 1
    elif op.startswith("s_cbranch") or op.startswith("s_branch"):
        cur[4].append(op.replace("s_cbranch_", "c_").replace("s_branch", "b") + ">" + m.group(2).strip())
    elif op.startswith("s_"):
        cur[2] += 1
    elif op.startswith(("ds_", "global_", "buffer_", "scratch_", "flat_")):
        cur[3] += 1
blocks.append(cur)
for b in blocks:
    print("%-12s L%-5d V%-4d S%-4d M%-3d %s" % (b[0], b[5], b[1], b[2], b[3], " ".join(b[4])))
