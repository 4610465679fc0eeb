"""one rank of tests/test_gpu_sharding_product.py: the flow bench.py uses for N > 1 with the PRODUCT on
every rank - shard of one alignment, partition on the harness's stream (pll_gpu_set_stream), log-likelihood
left on the device (pll_gpu_edge_loglikelihood_async), one all-reduce. gloo stands in for RCCL (two ranks
share the one GPU of the test box, which RCCL refuses), so the device value takes one D2H copy first."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "libpll-2_amd"))
sys.path.insert(0, ROOT)


def main():
    rank, world, port, kwjson = int(sys.argv[1]), int(sys.argv[2]), sys.argv[3], sys.argv[4]
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=port, PLL_AMD_DEVICE="0")
    import numpy as np
    import torch
    import torch.distributed as dist
    from pllamd import api, driver, sharding, workload as W
    dist.init_process_group("gloo", rank=rank, world_size=world)
    kw = json.loads(kwjson)
    case = W.make_case("full", **kw)
    sub = sharding.shard_case(case, rank, world)
    lib = api.PllLib()
    stream = torch.cuda.Stream()
    red = torch.zeros(2, dtype=torch.float64, device="cuda:0")
    fi = np.ascontiguousarray(sub.freqs_indices, dtype=np.uint32)
    vals = []
    with driver.Session(lib, sub, api.ARCH_AVX2) as s, torch.cuda.stream(stream):
        assert lib.pll_gpu_set_stream(s.p, stream.cuda_stream)
        e = sub.edges[0]
        for it in range(3):
            s.update_partials()
            assert lib.pll_gpu_edge_loglikelihood_async(s.p, e[0], e[1], e[2], e[3], e[4], api.uptr(fi), red.data_ptr())
            host = red.cpu()  # stream-ordered behind the evaluation
            t = host[:1].clone()
            dist.all_reduce(t)
            vals.append(float(t.item()))
        sync_v, _ = s.edge_lnl(e, persite=False)
    dist.barrier()
    dist.destroy_process_group()
    print(json.dumps(dict(rank=rank, totals=vals, own=sync_v, sites=sub.sites)))


if __name__ == "__main__":
    main()
