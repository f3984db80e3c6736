cd /root/repo
run() { tag=$1; shift; env "$@" python bench.py --steps 60 --warmup 10 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.loads(sys.stdin.read()); print('$tag', round(d['ms_per_step'],4), d['stage_ms'], d['config'].get('emitted_instances'))"; }
run base A=1
run d8 VCR_SORT_DIGIT_BITS=8
run d8_g512 VCR_SORT_DIGIT_BITS=8 VCR_SIDE_GRID=512
run d8_g384 VCR_SORT_DIGIT_BITS=8 VCR_SIDE_GRID=384
run d8_g128 VCR_SORT_DIGIT_BITS=8 VCR_SIDE_GRID=128
run d8_sortstream VCR_SORT_DIGIT_BITS=8 VCR_SORT_STREAM_MIN=0
run sparse0 VCR_SORT_DIGIT_BITS=8 VCR_X=1
VCR_LIB=$PWD/vcr_gaus_amd/libvcr_hithist.so python profiles/hit_histogram.py metric_1m_1080p > gpurun_out/r3_hit_histogram_metric.txt 2>&1; head -12 gpurun_out/r3_hit_histogram_metric.txt
VCR_LIB=$PWD/vcr_gaus_amd/libvcr_hithist.so python profiles/hit_histogram.py dense_1m_1080p > gpurun_out/r3_hit_histogram_dense.txt 2>&1; head -8 gpurun_out/r3_hit_histogram_dense.txt
