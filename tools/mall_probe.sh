set -e
mkdir -p gpurun_out/mall
for f in 16 32 48 64 128 256; do
  python bench.py --frames $f --steps 60 --warmup 5 --no-extra --no-cpu-baseline --no-verify > gpurun_out/mall/f$f.json 2> gpurun_out/mall/f$f.err
  echo done $f
done
