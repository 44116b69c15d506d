# whole steps of the in-tree build against a build with another instruction-scheduling strategy: usage ab_sched.sh <lib>
L=$1
echo "== 2-D fast 4096^2" && bash scripts/exp/ab_step.sh $L &&
echo "== 3-D fast 512^3" && bash scripts/exp/ab_step.sh $L --nx 512 --ny 512 --nz 512 --steps 40 --warmup 5 &&
echo "== 2-D strict 4096^2" && bash scripts/exp/ab_strict_step.sh $L &&
echo "== 3-D strict 256^3" && bash scripts/exp/ab_strict_step.sh $L --nx 256 --ny 256 --nz 256 --steps 40 --warmup 5
