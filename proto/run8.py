import run6, sys
for af, ah in [(0,0),(5,0),(20,0),(20,1),(20,2),(20,5),(50,2),(5,2)]:
    kw=dict(admm_first=af, admm_hot=ah, sp=1e-12, max_rounds=200)
    r = run6.run(kw, ['0','1','2','3'])
    print(af, ah, [(x[2], x[3], x[4], x[5]) for x in r])
