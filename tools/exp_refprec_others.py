import sys, json, argparse
sys.path.insert(0, '/root/repo')
import torch, bench
dev = torch.device("cuda", 0)
env = {"world": 1, "rank": 0, "local_rank": 0, "use_dist": False, "dev": dev}
print(json.dumps(bench.reference_precision_other_configs(None, env), indent=1))
