# usage: tools/sample_clocks.sh <outfile> <command...>   — samples rocm-smi (sclk, mclk, power, temperature) every ~0.25 s while the command runs
out=$1; shift
( while true; do date +%s.%N; /opt/rocm/bin/rocm-smi --showclocks --showpower --showtemp --showuse 2>/dev/null | grep -E "sclk|mclk|Power|Temperature \(Sensor junction|GPU use" ; sleep 0.2; done ) > "$out" 2>&1 &
spid=$!
"$@"
rc=$?
kill $spid 2>/dev/null
wait $spid 2>/dev/null
exit $rc
