#!/bin/bash
# usage: tools/asmstat.sh <file.hip> <kernel-name-regex>   (register / spill counts of the kernels of one source file)
F=$1; PAT=$2
/opt/rocm/bin/hipcc -O3 -std=c++17 -fPIC --offload-arch=gfx950 -S --cuda-device-only /root/repo/scri_amd/csrc/$F -o /tmp/asmstat.s 2>&1 | grep -E "error" -A5
python3 - "$PAT" <<'PY'
import re,sys
s=open('/tmp/asmstat.s').read()
for m in re.finditer(r'\.name:\s+(\S+)', s):
    if re.search(sys.argv[1], m.group(1)):
        seg=s[m.start()-1500:m.start()+1500]
        print(m.group(1)[:70], re.findall(r'\.(vgpr_count|vgpr_spill_count|group_segment_fixed_size):\s+(\d+)', seg))
PY
