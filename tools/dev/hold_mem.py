import torch, time, sys
gb = int(sys.argv[1])
x = [torch.empty(1 << 30, dtype=torch.uint8, device="cuda") for _ in range(gb)]
for t in x: t.zero_()
torch.cuda.synchronize()
print("holding", gb, "GB", flush=True)
time.sleep(float(sys.argv[2]))
