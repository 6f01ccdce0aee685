for th in 88888 88884 88844 48888 44444 84448; do
  for p in f32 split_f16; do
    echo "TH=$th $p"; SRHIP_TH=$th timeout 120 python bench.py --steps 20 --no-cpu-baseline --precision $p 2>/dev/null | python -c "
import sys,json
d=json.loads(sys.stdin.read().strip().splitlines()[-1])
print(' ms', d['ms_per_step'], [round(s['ms'],3) for s in d['stages']])"
  done
done
