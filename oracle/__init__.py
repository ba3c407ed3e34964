# oracle/ -- CPU restatement of the reference hot path. TEST INFRASTRUCTURE ONLY.
