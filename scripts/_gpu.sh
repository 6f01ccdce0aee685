set -u
export TMPDIR=/tmp
for v in 8 12 16 32 0; do
echo "== SRHIP_CONV0_PER_CU=$v"
SRHIP_CONV0_PER_CU=$v timeout 100 python scripts/band_profile.py f32 9 2>/dev/null | python -c "
import json,sys
for l in sys.stdin:
    d=json.loads(l)
    if 'full_frame' in d: print('full', d['full_frame']['wall_ms'], d['full_frame']['stage_ms'][0]); continue
    if d['th']=='mixed': print(d['ways'], d['wall_ms'], d['stage_ms'][0])
"
SRHIP_CONV0_PER_CU=$v timeout 60 python scripts/run_once.py f32 1080x1920 30 >/dev/null 2>&1
done
