import torch, time
torch.cuda.init(); torch.zeros(1, device='cuda'); torch.cuda.synchronize()
for gb in (0.5, 2, 8, 24):
    t=time.perf_counter(); x=torch.empty(int(gb*2**30), dtype=torch.uint8, device='cuda'); torch.cuda.synchronize(); t1=time.perf_counter()-t
    print(f"malloc {gb} GB: {t1*1e3:.1f} ms", flush=True)
    del x
torch.cuda.empty_cache()
t=time.perf_counter()
xs=[torch.empty(int(0.8*2**30), dtype=torch.uint8, device='cuda') for _ in range(30)]
torch.cuda.synchronize(); print(f"30 x 0.8 GB: {(time.perf_counter()-t)*1e3:.1f} ms")
del xs
t=time.perf_counter()
xs=[torch.empty(int(0.8*2**30), dtype=torch.uint8, device='cuda') for _ in range(30)]
torch.cuda.synchronize(); print(f"again (cached): {(time.perf_counter()-t)*1e3:.1f} ms")
del xs; torch.cuda.empty_cache()
t=time.perf_counter(); big=torch.empty(int(24*2**30), dtype=torch.uint8, device='cuda'); del big
xs=[torch.empty(int(0.8*2**30), dtype=torch.uint8, device='cuda') for _ in range(28)]
torch.cuda.synchronize(); print(f"24 GB pool then 28 x 0.8 GB carved: {(time.perf_counter()-t)*1e3:.1f} ms; reserved {torch.cuda.memory_reserved()/2**30:.1f} GB")
