#!/bin/bash
# VALU issue share of every kernel of some bench groups:  bash tools/valu_busy.sh group1 group2 ...
for G in "$@"; do
  bash $GRAFT_REPO_ROOT/tools/pmc_piece.sh group:$G "SQ_WAVES SQ_INSTS_VALU GRBM_GUI_ACTIVE SQ_INSTS_SALU SQ_INSTS_LDS SQ_LDS_BANK_CONFLICT" > /tmp/vb.txt
  python3 - <<'PY'
import re
cur=None; d={}
for line in open('/tmp/vb.txt'):
    if not line.startswith(' '):
        cur=line.strip()[:60]; d[cur]={}
    else:
        k,v=line.split()[:2]; d[cur][k]=float(v)
for k,v in d.items():
    if 'GRBM_GUI_ACTIVE' not in v: continue
    cyc=v['GRBM_GUI_ACTIVE']/8
    print(f"{k:60s} {cyc/2400:7.1f} us  VALU/wave {v['SQ_INSTS_VALU']/max(v['SQ_WAVES'],1):7.0f}  VALU busy {100*v['SQ_INSTS_VALU']*4/1024/cyc:5.1f}%  SALU/wave {v['SQ_INSTS_SALU']/max(v['SQ_WAVES'],1):6.0f}  LDS conflict cyc/launch {v['SQ_LDS_BANK_CONFLICT']:.0f}")
PY
done
