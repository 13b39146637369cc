"""stub"""
chemical_symbols = ["X"] + ["E%d" % i for i in range(1, 119)]
atomic_numbers = {s: i for i, s in enumerate(chemical_symbols)}
atomic_masses = [1.0] * 119
