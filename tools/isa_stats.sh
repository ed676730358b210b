#!/bin/bash
# Static facts about the product kernel <CULL=1,COUNT=0,LDS=0>: instruction mix, registers, spills.
# usage: [ISA_HEAVY=1] tools/isa_stats.sh [extra hipcc flags]     (ISA_HEAVY=1: the many-primitive variant <..., HEAVY = true>; default the light one; SKYLOD = false)
cd "$(dirname "$0")/.."
OUT=${ISA_OUT:-/tmp/rt_kernel_isa.s}
cfg() { sed -n "s/^$1[ \t]*?=[ \t]*//p" raytracing_opengl_amd/kernel_build.cfg; }   # the product's own configuration
/opt/rocm/bin/hipcc -DRT_WAVES_PER_EU=$(cfg WAVES_PER_EU) -DRT_WPE_HEAVY=$(cfg WPE_HEAVY) --offload-arch=gfx950 $(cfg KERNEL_FLAGS) -Iinclude -Iraytracing_opengl_amd/csrc "$@" -S --cuda-device-only -o $OUT raytracing_opengl_amd/csrc/rt_kernel.hip 2>&1 | grep -v "warning\|^$"
python3 - "$OUT" <<'PY'
import re,sys
txt=open(sys.argv[1]).read()
import os
m=[x for x in re.finditer(r'rt_trace_kernelILb1ELb0ELb0ELi\d+ELb([01])ELb0EEEv14RtLaunchParams:', txt) if x.group(1) == os.environ.get('ISA_HEAVY', '0')]
i=m[0].start()
j=txt.find('.end_amdhsa_kernel',i)
body=txt[i:j]
ins=[l.strip() for l in body.split('\n') if l.startswith('\t') and not l.strip().startswith(('.',';'))]
valu=[l for l in ins if l.startswith('v_')]
c=lambda p: sum(l.startswith(p) for l in ins)
print('instrs',len(ins),'ds',c('ds_'),'valu',len(valu),'pk',c('v_pk_'),'writelane',c('v_writelane'),'readlane',c('v_readlane'),'s_load',c('s_load'),'scratch',c('scratch_'), 'div',c('v_div_fixup'))
print(re.findall(r'; (?:NumVgprs|ScratchSize|NumSgprs|Occupancy): \d+', txt[j:j+4000])[:4])
PY
