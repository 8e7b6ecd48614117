set -e
for t in "4=1,5=0" "4=1,5=1" "4=4,5=1" "4=8,5=0" "4=8,5=1" "4=2,5=1" "4=16,5=1"; do
  echo "== $t"
  timeout -k 10 120 python bench.py --grid 150 --steps 2 --warmup 1 --no-cpu --tune "$t" | tail -1
done
