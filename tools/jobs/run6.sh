for pr in 1 0 1 0; do
  VQA_SIDE_PRIORITY=$pr python bench.py --steps 10 --warmup 3 --no-cpu-baseline --stream-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('fp32 prio=$pr', d['ms_per_step'], d['value'])"
done
for pr in 1 0; do
  VQA_SIDE_PRIORITY=$pr python bench.py --batch 1024 --tokens 30 --answers 3000 --steps 4 --warmup 2 --no-cpu-baseline --stream-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('stress prio=$pr', d['ms_per_step'], d['value'])"
  VQA_SIDE_PRIORITY=$pr python bench.py --dtype bf16 --batch 512 --size 448 --steps 4 --warmup 2 --no-cpu-baseline --stream-steps 0 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('bf16 prio=$pr', d['ms_per_step'], d['value'])"
done
python -m pytest tests -m gpu -x -q -k "lstm" 2>&1 | tail -2
