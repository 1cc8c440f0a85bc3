#!/bin/bash
# A/B of prefilter kernel variants (libraries built by build_variant.sh): usage tools/abtest/pf_ab.sh
for lib in default tools/abtest/e_maxilp.so default; do
  if [ "$lib" = default ]; then unset RMDF_LIB; else export RMDF_LIB=$PWD/$lib; fi
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline --no-animated 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read()); s=d['secondary']['config5_lobe_prefilter_256x128']
print('$lib', {k:v['kernel_ms'] for k,v in s['per_power'].items()}, 'four', s['four_powers_concurrent_host_in_out_ms'], 'scene1', d['secondary']['scene1_detest_1280x720_m128']['kernel_ms_avg'], 'scene3', d['secondary']['scene3_mbgeneral_1280x720_m128']['kernel_ms_avg'])"
done
