set -e
for t in "2=256,7=1024" "2=128,7=512,8=2048" "2=256,7=1024,8=4096" "2=192,7=768" "2=320,7=1280" "2=256,7=768" "2=384,7=1536" "2=128,7=1024" "2=256,7=1280"; do
  echo "== $t"
  timeout -k 10 120 python bench.py --grid 150 --steps 2 --warmup 1 --no-cpu --tune "$t" | python -c "
import sys, json
d = json.loads(sys.stdin.readlines()[-1])
print(d['value'], d['ms_per_step'], d['config']['kernel_ms'], d['roofline']['achieved'], d['roofline']['launches'])"
done
