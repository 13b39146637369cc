"""stub"""
all_changes = ["positions", "numbers", "cell", "pbc"]
