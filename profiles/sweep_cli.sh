#!/bin/bash
# BASELINE configs[3] on one GPU: the 715-768 MHz sweep (531 carriers, n_f = 35) through the Python sweep driver
# (buffers resident in HBM before timing) and through the C++ CLI on 531 pre-written capbuf_NNNN.it files.
#   bash profiles/sweep_cli.sh <tag>
TAG=${1:-sweep}; OUT=$PWD/gpurun_out/$TAG; mkdir -p $OUT
D=/tmp/sweep_it; rm -rf $D
python tools/sweep_cellsearch.py -s 715e6 -e 768e6 --json --write-it $D > $OUT/sweep_715_768_n1.json 2> $OUT/sweep.err
tail -c 400 $OUT/sweep_715_768_n1.json
make -C host -s
t0=$(date +%s.%N)
./host/CellSearch -s 715e6 -e 768e6 -l -d $D -b > $OUT/cli_sweep.txt 2> $OUT/cli_sweep.err
t1=$(date +%s.%N)
./host/CellSearch -s 715e6 -e 768e6 -l -d $D -b > /dev/null 2>&1          # second run: the 1.3 GB of .it files are in the page cache
t2=$(date +%s.%N)
python3 -c "print('CellSearch -s 715e6 -e 768e6 -l on 531 capbuf_NNNN.it files (2.46 MB each, complex<double>): %.2f s wall first run, %.2f s second run' % ($t1 - $t0, $t2 - $t1))" | tee $OUT/cli_time.txt
tail -6 $OUT/cli_sweep.txt
sed -n '/Detected the following cells/,$p' $OUT/cli_sweep.txt > $OUT/cli_sweep_table.txt
rm -rf $D
