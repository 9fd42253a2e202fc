#!/bin/bash
# Sample the GPU's package power (hwmon power1_average, microwatts) and shader clock every 50 ms while a command runs:
#   tools/r5/power_sample.sh <label> <command ...>   -> one line "label: n samples, W mean / max, sclk MHz mean" + the command's output
label=$1; shift
HW=$(ls -d /sys/class/drm/card*/device/hwmon/hwmon* 2>/dev/null | head -1)
PW=""
for f in power1_average power1_input; do [ -r "$HW/$f" ] && PW="$HW/$f" && break; done
CLK=$(ls /sys/class/drm/card*/device/hwmon/hwmon*/freq1_input 2>/dev/null | head -1)
TMP=$(mktemp)
"$@" > $TMP.out 2>&1 &
pid=$!
sleep 0.35   # (skip the start-up)
while kill -0 $pid 2>/dev/null; do
  p=0; c=0
  [ -n "$PW" ] && p=$(cat $PW 2>/dev/null)
  [ -n "$CLK" ] && c=$(cat $CLK 2>/dev/null)
  echo "$p $c" >> $TMP
  sleep 0.05
done
wait $pid
cat $TMP.out
if [ -s $TMP ]; then
  med=$(awk '{print $1/1e6}' $TMP | sort -n | awk '{a[NR]=$1} END {print a[int((NR+1)/2)]}')
  awk -v l="$label" -v med="$med" '{n++; w=$1/1e6; s+=w; if (w>m) m=w; c+=$2/1e6} END {printf "POWER %s: %d samples, %.0f W mean, %.0f W median, %.0f W max, sclk %.0f MHz mean (hwmon)\n", l, n, s/n, med, m, c/n}' $TMP
else
  echo "POWER $label: no hwmon power file ($HW)"
fi
rm -f $TMP $TMP.out
