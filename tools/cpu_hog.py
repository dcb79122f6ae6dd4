"""n busy processes for s seconds (a noisy neighbour on the box's host cores): python tools/cpu_hog.py n s"""
import multiprocessing as mp, sys, time


def burn(seconds):
    t0 = time.time()
    x = 0
    while time.time() - t0 < seconds:
        for i in range(100000):
            x += i * i
    return x


if __name__ == "__main__":
    n, s = int(sys.argv[1]), float(sys.argv[2])
    ps = [mp.Process(target=burn, args=(s,)) for _ in range(n)]
    [p.start() for p in ps]
    [p.join() for p in ps]
