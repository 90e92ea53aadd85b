import run6, sys
for af, ah, sp in [(10,0,1e-8),(10,2,1e-8),(20,2,1e-8),(10,0,1e-10)]:
    kw=dict(admm_first=af, admm_hot=ah, sp=sp, max_rounds=200)
    print(af, ah, sp, run6.run(kw, ['circle','exdata']))
