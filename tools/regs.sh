#!/bin/bash
# usage: tools/regs.sh DEGREE GEOM [extra -D flags]  -- VGPR / scratch of the k_block_op variants of one degree
deg=$1; geom=$2; shift; shift
mkdir -p /tmp/regchk && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -munsafe-fp-atomics -DFUS_DEV_BUILD -DFUS_DEV_DEGREE=$deg -DFUS_TU_DEGREE=$deg "$@" --cuda-device-only -S $(dirname $0)/../fenicsx-fus_amd/csrc/fusmi.hip -o /tmp/regchk/p.s 2>/dev/null
python3 - $geom <<'PY'
import re,sys
s=open('/tmp/regchk/p.s').read()
for m in re.finditer(r'\.amdhsa_kernel (\S+)(.*?)\.end_amdhsa_kernel', s, re.S):
    name,body=m.group(1),m.group(2)
    t=re.search(r'k_block_opI(\w)Li(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)ELi(\d)E',name)
    if not t or t.group(7)!=sys.argv[1]: continue
    v=re.search(r'\.amdhsa_next_free_vgpr (\d+)',body).group(1)
    sp=re.search(r'\.amdhsa_private_segment_fixed_size (\d+)',body).group(1)
    print('T=%s P=%s OP=%s ATOMIC=%s STAGE=%s NF=%s GEOM=%s TD=%s'%t.groups(),'vgpr',v,'scratch',sp)
PY
