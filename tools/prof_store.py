import cProfile, pstats, sys, os, numpy as np, torch
sys.path.insert(0, os.getcwd())
from gcnn_cut_selector_amd import synthetic
from gcnn_cut_selector_amd.model import GCNN
from gcnn_cut_selector_amd.store import SampleStore
from gcnn_cut_selector_amd.trainer import Adam, process
dev = torch.device("cuda", 0)
samples = [synthetic.make_sample("setcov", i) for i in range(64)]
store = SampleStore.from_samples(samples, dev)
m = GCNN(device=dev); opt = Adam(learning_rate=lambda: 1e-3)
ids = np.random.default_rng(0).choice(64, 100 * 32)
fr = np.array([.25, .5, .75, 1.0])
process(m, store.batches(ids[:64], 32), fr, None, opt); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
process(m, store.batches(ids, 32), fr, None, opt); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats("cumulative").print_stats(28)
