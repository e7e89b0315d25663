import sys, os
R = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, R); sys.path.insert(0, os.path.join(R, "tests"))
import gpu_checks as G
for i in range(2):
    print("overlap", G.check_transducer_branch_overlap())
    print("encdec", G.check_encdec_deferred_matches_immediate())
print("joint", G.check_joint_wgrad())
