"""Development aid: instruction mix of a barrier-structured kernel, segment by segment (a segment = the code between two
consecutive s_barrier instructions in program order), from a hipcc -save-temps .s file: MFMA / other vector / transcendental /
LDS / global / scratch (spill) instructions and s_waitcnt.
    python tools/isa_scratch_map.py file.s KERNEL_NAME_SUBSTRING"""
import re
import sys

path, key = sys.argv[1], sys.argv[2]
lines = open(path).read().splitlines()
start = next(i for i, l in enumerate(lines) if l.startswith("_Z") and key in l and ":" in l)
end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
TRANS = ("v_exp", "v_log", "v_rcp", "v_rsq", "v_sqrt", "v_sin", "v_cos")
keys = ["mfma", "valu", "trans", "f64", "ds_rd", "ds_wr", "glob_ld", "glob_st", "scr_ld", "scr_st", "waitcnt", "salu", "branch"]
cnt = dict.fromkeys(keys, 0)
seg = 0
print("kernel lines", start, end)
print("seg  " + " ".join(f"{k:>7s}" for k in keys))


def flush():
    global seg
    print(f"{seg:3d}  " + " ".join(f"{cnt[k]:7d}" for k in keys))
    seg += 1
    for k in keys:
        cnt[k] = 0


for l in lines[start:end]:
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    op = t.split()[0]
    if op.startswith("s_barrier"):
        flush()
    elif op.startswith("v_mfma"):
        cnt["mfma"] += 1
    elif op.startswith(TRANS):
        cnt["trans"] += 1
    elif op.startswith("v_") and op.endswith("_f64"):
        cnt["f64"] += 1
    elif op.startswith("v_"):
        cnt["valu"] += 1
    elif op.startswith("ds_read") or op.startswith("ds_load"):
        cnt["ds_rd"] += 1
    elif op.startswith("ds_"):
        cnt["ds_wr"] += 1
    elif op.startswith("scratch_load"):
        cnt["scr_ld"] += 1
    elif op.startswith("scratch_store"):
        cnt["scr_st"] += 1
    elif op.startswith(("global_load", "flat_load", "buffer_load")):
        cnt["glob_ld"] += 1
    elif op.startswith(("global_store", "flat_store", "buffer_store", "global_atomic", "flat_atomic")):
        cnt["glob_st"] += 1
    elif op.startswith("s_waitcnt"):
        cnt["waitcnt"] += 1
    elif op.startswith(("s_cbranch", "s_branch")):
        cnt["branch"] += 1
    elif op.startswith("s_"):
        cnt["salu"] += 1
flush()
