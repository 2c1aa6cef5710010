cd $GRAFT_REPO_ROOT
echo "== world=1 via torch.distributed.run"
python -m torch.distributed.run --nnodes=1 --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29511 bench.py --gpus 1 --steps 5 --warmup 2 --no-cpu-baseline 2>&1 | grep -E '^\{|Error|error' | cut -c1-400
echo "== world=2 on ONE gpu (expected to be refused by RCCL; informational)"
MALLIE_FORCE_DEVICE0=1 timeout 120 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29512 bench.py --gpus 2 --steps 3 --warmup 1 2>&1 | grep -E '^\{|Error|error|Duplicate|invalid' | head -8 | cut -c1-300
