cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for form in new old; do
  if [ $form = old ]; then export HDM_GRAM_QUEUE=0 HDM_GRAM_KSTAGES=1563 HDM_NSPLIT=80; fi
  rocprofv3 --pmc TCC_EA0_RDREQ_LEVEL_sum TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_DRAM_sum GRBM_GUI_ACTIVE --kernel-trace --output-format csv -d $R/gpurun_out/lat_$form -- python3 $R/bench.py --m 8000 --steps 1 --warmup 0 --no-cpu > $R/gpurun_out/lat_$form.json 2> $R/gpurun_out/lat_$form.err
  python3 $R/tools/prof_summary.py $R/gpurun_out/lat_$form "$form" 2>&1 | grep -E "persist_kernel<true" | cut -c1-400
  rm -rf $R/gpurun_out/lat_$form
done
