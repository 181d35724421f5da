"""placeholder -- filled in below"""
def gen_code(*a, **k):
  raise NotImplementedError
class EKF_sym: pass
class BatchedEKF: pass
