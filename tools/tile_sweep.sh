set -e
for t in "" "2=1024" "2=768" "6=1" "0=128,1=64" "0=256,1=128"; do
  echo "== $t"
  timeout -k 10 120 python bench.py --grid 150 --steps 2 --warmup 1 --no-cpu --tune "$t" | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['kernel_ms'], d['roofline']['achieved'], d['roofline']['launches'])"
done
