out=/root/repo/gpurun_out/r02_final; mkdir -p $out
cd /tmp; export TMPDIR=/tmp
rm -rf $out/stencils $out/stencils_fetch $out/stencils_write
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stencils -o s -- python /root/repo/tools/bench_stencils.py > $out/bench_stencils.txt 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $out/stencils_fetch -o p -- python /root/repo/tools/bench_stencils.py > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $out/stencils_write -o p -- python /root/repo/tools/bench_stencils.py > /dev/null 2>&1
grep -v '^[WE]2026' $out/bench_stencils.txt
