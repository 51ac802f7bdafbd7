# samples rocm-smi while a load runs: usage scratch/power_probe.sh <out> -- command...
out=$1; shift; shift
"$@" > $out.load 2>&1 &
pid=$!
: > $out
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showclocks 2>/dev/null | grep -E "sclk" | tr '\n' ' ' >> $out
  echo >> $out
  sleep 0.3
done
wait $pid
