# round 6: the small configurations (128^3, 3.9k-leaf plume) under the existing option words -- is any non-default form faster there?
cd $GRAFT_REPO_ROOT
O=gpurun_out/r06m; mkdir -p $O
for rep in 1 2; do
for c in 128 plume; do
for o in - divergence=zpair divergence=coalesced sor_block_lb=1 schedule=linear; do
	timeout 120 python profiles/micro/bench_with_options.py $o --config $c --no-cpu-baseline --steps 40 >> $O/small.txt 2>> $O/err.txt
done; done; done
cat $O/small.txt
