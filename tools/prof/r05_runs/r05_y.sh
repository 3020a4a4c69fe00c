# round 5, pass y: the checksum beside the emission as well (the trailer written by a kernel of its own behind it), ZH_TRAILER_LATE=1 against 0
export TMPDIR=/tmp
O=gpurun_out; mkdir -p $O
ZH_TRAILER_LATE=1 timeout 600 python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "identical or fixtures or ragged or config2 or plan" 2>&1 | tail -2
run() { env $1 timeout 400 python bench.py --steps 5 --warmup 2 --no-cpu-baseline --no-configs --no-parity-sample "${@:2}" 2>/dev/null | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$1', d['value'], d.get('value_parallel_parse'), d['compress_GiBps'], {k:round(v,3) for k,v in d['kernels_ms'].items() if k in ('zh_huffman_kernel','zh_emit_kernel','zh_checksum_pieces_kernel','zh_layout_kernel','zh_trailer_kernel')}, {k:round(v,3) for k,v in ((d.get('parallel_parse') or {}).get('kernels_ms') or {}).items() if k in ('zh_huffman_kernel','zh_emit_kernel','zh_checksum_pieces_kernel','zh_trailer_kernel')})"; }
echo "== full"; for r in 1 2; do run ZH_TRAILER_LATE=1; run ZH_TRAILER_LATE=0; done
echo "== share512"; run ZH_TRAILER_LATE=1 --buffers 512; run ZH_TRAILER_LATE=0 --buffers 512
echo "== c2"; run ZH_TRAILER_LATE=1 --buffers 1024 --size 65536 --compress-only --steps 20 --no-parallel-parse; run ZH_TRAILER_LATE=0 --buffers 1024 --size 65536 --compress-only --steps 20 --no-parallel-parse
