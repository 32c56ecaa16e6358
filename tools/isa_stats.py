#!/usr/bin/env python3
"""Per-kernel instruction mix of a device assembly listing (hipcc --cuda-device-only -S): static counts of VALU / SALU / LDS /
VMEM / scratch instructions, packed fp32, readlane, compares, and the register / scratch / LDS footprint.
usage: isa_stats.py apd.s [name-filter ...]"""
import re
import sys

s = open(sys.argv[1]).read()
filt = sys.argv[2:]
meta = {}
for blk in re.findall(r"- \.agpr_count:.*?\.wavefront_size:\s+\d+", s, flags=re.S):
    nm = re.search(r"\.name:\s+(\S+)", blk).group(1)
    g = lambda k: int(re.search(r"\.%s:\s+(\d+)" % k, blk).group(1))
    meta[nm] = (g("vgpr_count"), g("sgpr_count"), g("private_segment_fixed_size"), g("group_segment_fixed_size"))
for m in re.finditer(r"^(\S+):\s*; @\S+\n(.*?)^\.Lfunc_end\d+:", s, flags=re.S | re.M):
    name, body = m.group(1), m.group(2)
    if filt and not any(f in name for f in filt):
        continue
    ins = [l.split()[0] for l in body.split("\n") if l.startswith("\t") and not l.startswith("\t.") and not l.startswith("\t;") and l.strip()]
    c = lambda pat: sum(1 for i in ins if re.match(pat, i))
    vg, sg, scr, lds = meta.get(name, (0, 0, 0, 0))
    short = re.sub(r"^_ZN3apd\d+", "", name)[:28]
    print(f"{short:28s} instr {len(ins):5d}  valu {c(r'v_'):5d} (pk_f32 {c(r'v_pk_(mul|add|fma)_f32')}, readlane {c(r'v_readlane|v_readfirstlane')}, cmp {c(r'v_cmp')})  salu {c(r's_(?!waitcnt|nop|endpgm|barrier|load|buffer|store)'):5d} "
          f"(bcnt {c(r's_bcnt')}, branch {c(r's_cbranch')})  smem {c(r's_load|s_buffer_load')} lds {c(r'ds_')}  vmem {c(r'global_|buffer_|flat_')} scratch {c(r'scratch_')}  waitcnt {c(r's_waitcnt')} | vgpr {vg} sgpr {sg} scratch {scr} B lds {lds} B")
