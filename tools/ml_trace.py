import csv, glob, sys
f = glob.glob(sys.argv[1] + "/**/*_kernel_trace.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
sel = [(r["Kernel_Name"], int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Stream_Id", r.get("Queue_Id", "")), r["Grid_Size_X"] if "Grid_Size_X" in r else "") for r in rows]
t0 = sel[0][1]
# find eigen-pass region: first k_td_col
tri = [(n, s, e, q, g) for n, s, e, q, g in sel if "k_td_trail_tri<3, 0" in n or "k_td_trail_tri<3, 4" in n or "k_td_solve" in n or "k_td_col" in n]
print("n td kernels", len(tri))
first_col = next(i for i, x in enumerate(tri) if "k_td_col" in x[0])
# print the first 12 and, later, launches around the middle of 2nd half batch
def show(lst):
    for n, s, e, q, g in lst:
        print(f"{(s-t0)/1e6:10.3f} ms  dur {(e-s)/1e3:9.1f} us  q={q} grid={g}  {n[:40]}")
show(tri[first_col:first_col+14])
print("...")
# second half-batch start: find second occurrence where grid of tri<3,0 jumps back up
big = [i for i, x in enumerate(tri) if "trail_tri<3, 0" in x[0] and (x[2]-x[1]) > 400e3]
print("big read-only sweeps (>400us):", len(big))
for i in big[:6] + big[-6:]:
    n, s, e, q, g = tri[i]
    print(f"{(s-t0)/1e6:10.3f} ms  dur {(e-s)/1e3:9.1f} us  q={q} grid={g}")
sol = [x for x in tri if "k_td_solve" in x[0]]
for n, s, e, q, g in sol[:8]:
    print(f"solve {(s-t0)/1e6:10.3f} .. {(e-t0)/1e6:10.3f} ms q={q}")
