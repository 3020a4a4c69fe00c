# round 5, pass z: with the trailer behind the emission, the slots cleared on the checksum's stream ahead of it (ZH_CLEAR_LATE=1 against 0)
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
ZH_CLEAR_LATE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "identical or fixtures or ragged or plan" 2>&1 | tail -2
run() { env $1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('value_parallel_parse'), d['compress_GiBps'], {k:round(v,3) for k,v in d['kernels_ms'].items() if k in ('zh_huffman_kernel','zh_emit_kernel','zh_checksum_pieces_kernel','zh_layout_kernel','zh_trailer_kernel','memset_dst')}, {k:round(v,3) for k,v in ((d.get('parallel_parse') or {}).get('kernels_ms') or {}).items() if k in ('zh_huffman_kernel','zh_emit_kernel','zh_checksum_pieces_kernel','zh_trailer_kernel','zh_layout_kernel')})"; }
echo "== full"; for r in 1 2; do run ZH_CLEAR_LATE=1; run ZH_CLEAR_LATE=0; done
echo "== share512 (ZH_TRAILER_LATE=1 for both)"; export ZH_TRAILER_LATE=1; run ZH_CLEAR_LATE=1 --buffers 512; run ZH_CLEAR_LATE=0 --buffers 512
