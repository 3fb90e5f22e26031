module github.com/klauspost/compress/kcgpu

go 1.24

require github.com/klauspost/compress v1.18.0
