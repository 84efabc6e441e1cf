// Cameras.h -- ARenderer plug-in seam + Camera, mirroring src/Scene/Cameras/{ARenderer.h,
// PathTracingRenderer.{h,cpp},Camera.{h,cpp}} on top of the libcloudtrace C ABI.
#pragma once

#include "Exr.h"
#include <algorithm>
#include <cfloat>
#include <filesystem>

#include "Scene.h"

namespace DeepestScatter
{
    // What the reference passes around as optix::Program camera / optix::Buffer frameResultBuffer.
    struct CameraProgram
    {
        float3 eye{}, U{}, V{}, W{};
        bool valid = false;
    };

    class ARenderer                                                              // ARenderer.h:6-16
    {
    public:
        ARenderer() = default;
        virtual ~ARenderer() = default;

        virtual CameraProgram* getCamera() = 0;
        virtual void init() = 0;
        virtual void render(float* frameResultBuffer /* device pointer, W*H*4 floats, or nullptr */) = 0;
    };

    class PathTracingRenderer : public ARenderer                                // PathTracingRenderer.h:8-31
    {
    public:
        explicit PathTracingRenderer(std::shared_ptr<Context> context) : context(std::move(context)) {}
        ~PathTracingRenderer() override = default;

        CameraProgram* getCamera() override { return &camera; }

        void init() override                                                     // PathTracingRenderer.cpp:14-19
        {
            // all scene items have published their variables by now (installApp order, installers.cpp:32-35)
            CtScene& s = context->scene;
            s.density_host = context->density.data();
            s.mie_host = context->mie.data();
            s.chopped_mie_host = context->choppedMie.data();
            s.mie_count = (uint32_t)context->mie.size();
            s.estimator = estimator;
            context->create();
            context->density.clear();
            context->density.shrink_to_fit();
        }

        void render(float* frameResultBuffer) override                           // PathTracingRenderer.cpp:21-31
        {
            if (context->group) throw std::runtime_error("the two-launch loop (--unfused) renders on one GPU only");
            if (camera.valid)
            {
                context->setCamera(camera.eye.data(), camera.U.data(), camera.V.data(), camera.W.data());
                camera.valid = false;
            }
            Context::check(ct_render_subframe(context->handle, subframeId, frameResultBuffer), context->handle, "ct_render_subframe");
        }

        uint32_t subframeId = 0;        // context["subframeId"], set by Camera::render (Camera.cpp:191-192)
        int estimator = CT_EST_MARCH;   // CT_EST_DELTA: Woodcock tracking instead of the reference's fixed-step march
        inline static const std::string NAME = "PT";

    private:
        std::shared_ptr<Context> context;
        CameraProgram camera;
    };

    class Camera : public SceneItem                                              // Camera.h:16-103
    {
    public:
        struct Settings
        {
            Settings(uint32_t width, uint32_t height, std::filesystem::path outputFile)
                : width(width), height(height), outputFile(std::move(outputFile)) {}
            uint32_t width, height;
            std::filesystem::path outputFile;
        };

        Camera(std::shared_ptr<Settings> settings, std::shared_ptr<Context> context, std::shared_ptr<ARenderer> renderer)
            : width(settings->width), height(settings->height), outputFile(settings->outputFile),
              context(std::move(context)), renderer(std::move(renderer))
        {
            this->context->scene.width = width;
            this->context->scene.height = height;
        }

        void init() override                                                     // Camera.cpp:22-66
        {
            renderer->init();
            cameraEye = { 2.5f, -0.4f, 0.f };                                    // :37-39
            cameraLookat = { 0.f, 0.f, 0.f };
            cameraUp = { 0.f, 1.f, 0.f };
            updatePosition();
            reset();
        }

        void update() override { if (!isCompleted()) { updatePosition(); render(); } }   // :68-75

        void reset() override                                                    // :77-86
        {
            subframeId = 0;
            context->resetAccumulation();
            // headless: the test the reference makes before every update of 10 (:179, :232-268) runs on the device behind every
            // 10th subframe's accumulate kernel and freezes the image where the reference's loop would stop
            deviceStops = headless && fused && context->stopWhenConverged(subframesPerUpdate, 100);
        }

        bool isCompleted() override { return completed; }

        void setEye(const float3& eye) { cameraEye = eye; updatePosition(); reset(); }    // stands in for rotate(), :93-98
        void increaseExposure() { exposure *= 1.2f; }
        void decreaseExposure() { exposure /= 1.2f; }

        // How many subframes one render() call adds (10 in the reference, Camera.cpp:189) and when to stop
        // regardless of convergence (the reference runs until converged).
        uint32_t subframesPerUpdate = 10;
        uint32_t maxSubframes = 0;      // 0 = until isConverged()
        bool fused = true;              // one ct_render_accumulate per update instead of 2 launches per subframe
        bool headless = false;          // no display: enqueue the batches, read the buffers only where the reference saves
        bool completed = true;          // Camera.h:55

        bool deviceStops = false;       // the convergence test runs on the device (ct_set_stop_when_converged)

        uint32_t getSubframeId() const { return subframeId; }
        const std::vector<uint8_t>& getScreen() const { return screen; }

        void saveToDisk() const                                                  // :149-175
        {
            std::vector<float> mean((size_t)width * height * 4);
            context->downloadMean(mean.data(), mean.size() * sizeof(float));
            std::cout << mean[((size_t)width * height / 2 + width / 2) * 4] << std::endl;   // :163
            if (outputFile.extension() == ".pfm")
            {
                // RGB float32, bottom row first, which is the buffer's own row order (row 0 = bottom)
                std::ofstream f(outputFile, std::ios::binary);
                f << "PF\n" << width << " " << height << "\n-1.0\n";
                for (size_t i = 0; i < (size_t)width * height; i++) f.write(reinterpret_cast<const char*>(&mean[4 * i]), 3 * sizeof(float));
            }
            else
            {
                // EXR, channels R, G, B FLOAT, DECREASING_Y, pixel (x, y) = progressive[y * width + x] like the reference
                Exr::writeRgbFloat(outputFile.string(), width, height, mean.data());
            }
        }

    private:
        void updatePosition()                                                    // :100-134 (arcball path is UI-only)
        {
            const float hfov = 30.0f;
            const float aspectRatio = static_cast<float>(width) / static_cast<float>(height);
            CameraProgram* camera = renderer->getCamera();
            if (camera != nullptr)
            {
                ct_calculate_camera_variables(cameraEye.data(), cameraLookat.data(), cameraUp.data(), hfov, aspectRatio,
                                              camera->U.data(), camera->V.data(), camera->W.data());
                camera->eye = cameraEye;
                camera->valid = true;
            }
        }

        void render()                                                            // :177-230
        {
            // headless: nobody looks at the screen between saves, so batches are enqueued (a launch hands its
            // unfinished paths to the next one instead of ending with a tail) and the buffers are only read --
            // tonemap, save -- every 40 subframes, where the reference saves (:211-214).  The reference tests convergence
            // before every update of 10 subframes: the device does that behind every 10th subframe (deviceStops) and freezes
            // the image at the count where the reference's loop stops; the host learns of it at the next save point, takes
            // that count and that image, and is done -- the subframes enqueued meanwhile were rendered and dropped.  (A group
            // of GPUs tests its merged frame at the save points only and may stop up to 30 subframes later.)
            const bool look = !headless || subframeId % 40 == 0;
            if (deviceStops && look && subframeId >= 100)
            {
                uint64_t left = 0;
                const uint32_t at = context->convergedAt(left);   // (the tonemap of this save point has waited for everything)
                if (at != 0)
                {
                    std::cout << "Converged: " << (uint64_t)width * height - left << "/" << (uint64_t)width * height << " --- " << left << "left" << std::endl;
                    subframeId = at;
                    completed = true;
                    std::cout << "rendering subframe " << subframeId << std::endl;
                    saveToDisk();
                    return;
                }
            }
            if (!(look && !deviceStops && isConverged()) && !(maxSubframes && subframeId >= maxSubframes))
            {
                // headless: one batch up to the next point where anything is read (at 1024^2 a batch of 40 costs 15.4 ms where
                // four of 10 cost 18.5: a short launch has every pixel group in flight at once and misses L2 half again as
                // often); results do not depend on the batching
                uint32_t count = headless ? 40 - subframeId % 40 : subframesPerUpdate;
                if (maxSubframes) count = std::min(count, maxSubframes - subframeId);
                auto* pt = dynamic_cast<PathTracingRenderer*>(renderer.get());
                if (fused && pt != nullptr)
                {
                    CameraProgram* camera = renderer->getCamera();
                    if (camera->valid)
                    {
                        context->setCamera(camera->eye.data(), camera->U.data(), camera->V.data(), camera->W.data());
                        camera->valid = false;
                    }
                    context->renderAccumulate(subframeId + 1, count, headless);
                    subframeId += count;
                    std::cout << "rendering subframe " << subframeId << std::endl;
                }
                else
                {
                    for (uint32_t i = 0; i < count; i++)
                    {
                        subframeId++;
                        if (pt) pt->subframeId = subframeId;                      // context["subframeId"]->setUint, :192
                        std::cout << "rendering subframe " << subframeId << std::endl;
                        renderer->render(nullptr);                                // :195
                        Context::check(ct_accumulate(context->handle, subframeId, nullptr), context->handle, "ct_accumulate");   // :197-199
                    }
                }
                if (!headless || subframeId % 40 == 0)
                {
                    screen.resize((size_t)width * height * 4);
                    context->tonemap(exposure, screen.data());                    // :202-210
                }
                if (subframeId % 40 == 0) saveToDisk();                           // :211-214
            }
            else
            {
                if (deviceStops)
                {
                    // (the subframe limit came first -- or did it?  the image may have frozen since the last save point)
                    uint64_t left = 0;
                    context->wait();
                    const uint32_t at = context->convergedAt(left);
                    if (at != 0) subframeId = at;
                }
                completed = true;
                std::cout << "rendering subframe " << subframeId << std::endl;
                saveToDisk();
            }
        }

        bool isConverged()                                                       // :232-268 (evaluated on the device)
        {
            if (subframeId < 100) return false;
            uint64_t left = 0;
            const bool converged = context->isConverged(left);
            std::cout << "Converged: " << (uint64_t)width * height - left << "/" << (uint64_t)width * height << " --- " << left << "left" << std::endl;
            return converged;
        }

        uint32_t width, height;
        std::filesystem::path outputFile;
        std::shared_ptr<Context> context;
        std::shared_ptr<ARenderer> renderer;
        uint32_t subframeId = 0;
        float3 cameraUp{}, cameraLookat{}, cameraEye{};
        float exposure = 0.4f;                                                   // Camera.h:90
        std::vector<uint8_t> screen;
    };
}
