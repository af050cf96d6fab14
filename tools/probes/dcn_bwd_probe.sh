# kernel times of the DCNv3 kernels at InternImage-XL's four level geometries (tools/probes/dcn_bwd_one.py) from rocprofv3: the shipped library, and any
# A/B builds named on the command line (tools/_abl/libmtp_hip_<name>.so)
cd /tmp; export TMPDIR=/tmp
for n in base "$@"; do
  if [ $n = base ]; then L=""; else L="MTP_HIP_LIB=$GRAFT_REPO_ROOT/tools/_abl/libmtp_hip_$n.so"; fi
  env $L timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/dp$n -o t -- python $GRAFT_REPO_ROOT/tools/probes/dcn_bwd_one.py > /tmp/dp$n.log 2>&1
  f=$(find /tmp/dp$n -name t_kernel_trace.csv | head -1)
  python $GRAFT_REPO_ROOT/tools/probes/intern_by_grid.py $f dcnv3 --all | grep -v "last step" | sed "s/^/$n: /"
done
