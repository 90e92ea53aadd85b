import run6, sys
for sp in [1e-12, 1e-8, 1e-6, 1e-4, 1e-2]:
    kw=dict(admm_first=20, admm_hot=2, sp=sp, max_rounds=200)
    print(sp, run6.run(kw, sys.argv[1:]))
