# round 5, pass s: the output slots cleared on a third stream beside the code builder and the checksum (ZH_CLEAR_ASIDE=0: in front of the match finder, as before)
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "huffman or identical or fixtures or config2 or ragged or reruns or plan" > $O/r05_s_pytest_sub.log 2>&1; tail -2 $O/r05_s_pytest_sub.log
run() { env $1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('value_parallel_parse'), d['compress_GiBps'], {k:round(v,3) for k,v in d['kernels_ms'].items() if v > 0.5}, {k:round(v,3) for k,v in ((d.get('parallel_parse') or {}).get('kernels_ms') or {}).items() if k in ('zh_huffman_kernel','memset_dst','zh_checksum_pieces_kernel')})"; }
echo "== full"; for r in 1 2; do run ZH_CLEAR_ASIDE=1; run ZH_CLEAR_ASIDE=0; done
echo "== share512"; run ZH_CLEAR_ASIDE=1 --buffers 512; run ZH_CLEAR_ASIDE=0 --buffers 512
echo "== c2"; run ZH_CLEAR_ASIDE=1 --buffers 1024 --size 65536 --compress-only --steps 20 --no-parallel-parse; run ZH_CLEAR_ASIDE=0 --buffers 1024 --size 65536 --compress-only --steps 20 --no-parallel-parse
