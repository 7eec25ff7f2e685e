import cProfile, os, pstats, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import audiolm_pytorch_amd as A
dev = torch.device('cuda'); torch.manual_seed(0)
sem = A.SemanticTransformer(dim=1024, depth=6, num_semantic_tokens=500, flash_attn=True).to(dev)
sw = A.SemanticTransformerWrapper(transformer=sem, unique_consecutive=False).eval()
sw.generate(max_length=16, batch_size=1); torch.cuda.synchronize()
pr = cProfile.Profile(); pr.enable()
sw.generate(max_length=200, batch_size=1); torch.cuda.synchronize()
pr.disable()
pstats.Stats(pr).sort_stats('tottime').print_stats(28)
pstats.Stats(pr).sort_stats('cumulative').print_stats(30)
