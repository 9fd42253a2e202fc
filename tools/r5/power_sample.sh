#!/bin/bash
# Package power (rocm-smi --showpower, the visible GPU) sampled while a command runs:
#   tools/r5/power_sample.sh <label> <command ...>   -> the command's output, then "POWER label: n samples, W mean / median / max"
label=$1; shift
TMP=$(mktemp)
"$@" > $TMP.out 2>&1 &
pid=$!
sleep 1.0   # (skip start-up, warm-up and the first iterations)
while kill -0 $pid 2>/dev/null; do
  rocm-smi --showpower 2>/dev/null | grep -i "power (W)" | head -1 | sed 's/.*: *//' >> $TMP
done
wait $pid
cat $TMP.out
if [ -s $TMP ]; then
  n=$(wc -l < $TMP)
  # (the last sample may fall behind the end of the run: drop it when there are enough)
  [ $n -gt 3 ] && head -n $((n - 1)) $TMP > $TMP.s || cp $TMP $TMP.s
  med=$(sort -n $TMP.s | awk '{a[NR]=$1} END {print a[int((NR+1)/2)]}')
  awk -v l="$label" -v med="$med" '{n++; s+=$1; if ($1>m) m=$1} END {printf "POWER %s: %d samples, %.0f W mean, %.0f W median, %.0f W max (rocm-smi --showpower)\n", l, n, s/n, med, m}' $TMP.s
else
  echo "POWER $label: rocm-smi gave no power reading"
fi
rm -f $TMP $TMP.out $TMP.s
