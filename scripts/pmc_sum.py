import csv, sys, collections
acc = collections.defaultdict(lambda: [0.0, 0])
for row in csv.DictReader(open(sys.argv[1])):
    if "igemm_kernel" in row["Kernel_Name"]:
        k = (row["Kernel_Name"].split("<")[1].split(">")[0], row["Counter_Name"])
        acc[k][0] += float(row["Counter_Value"]); acc[k][1] += 1
for k, (v, n) in sorted(acc.items()):
    print(k, f"{v/n:.4g} per launch ({n})")
